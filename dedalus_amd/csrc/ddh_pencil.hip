// Batched pencil linear algebra for ALL pencils of a problem at once, gfx950.
//
// Layout.  A system vector is a real array [nrows][nx][ny]; a "cell" is one
// group of separable real-Fourier modes (mx, my) holding 2^nfourier real parts (cos/msin
// combinations).  Every translation-invariant real operator acts on the complex combinations
//     nfourier = 1:  P = a + i b                                         symbol lambda( kx)
//     nfourier = 2:  P = (cc - ss) + i (cs + sc)                          symbol lambda( kx, ky)
//                    Q = (cc + ss) + i (cs - sc)                          symbol lambda(-kx, ky)
// as an independent complex linear system, so one cell = S (1 or 2) complex "systems".  The cell
// index is the fastest-varying index of every array: thread g works on system g, all memory
// accesses of a wave are contiguous, and no gather/scatter pass exists at all (the reference's
// gather_inputs / scatter_inputs, core/subsystems.py:340-380, are the identity here).
//
// Matrices are term lists  A[row, col] += coef * kx^ex * ky^ey * [mx==0]^dx * [my==0]^dy  shared by
// all pencils (no per-pencil matrix values are ever stored or read for mat-vecs).
//
// The implicit solve is a bordered band LU with partial pivoting inside the band
// (replaces SuperLU, libraries/matsolvers.py:126-149):
//     columns = [interior variables, coupled-axis index outermost | border variables (taus)]
//     rows    = [interior equations, coupled-axis index outermost | border equations (BCs)]
// rows < n form a band (kl, ku); the nb border rows are dense and are eliminated without taking
// part in pivoting; systems whose band block is singular are flagged and solved through an
// explicit dense inverse supplied by the host (ddh_pencil_set_dense_inverse).
//
// Sweep kernels: one thread per system -- solve_forward_lean_kernel (real factors, two Fourier
// axes), solve_forward_kernel (everything else), solve_backward_kernel (guard-free for real factors) -- or 16 / 4 lanes
// per system for few systems (solve_*_coop_kernel).  Real factors are stored padded, pair-packed and, for problems
// symmetric under x <-> y, once per pair of transposed cells (LuDev::kpad, pk, pair).
#include "ddh_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace ddh {

constexpr int KLMAX = 16;
constexpr int NBMAX = 8;

struct PencilDev {
    int nf;      // separable real-Fourier axes: 0, 1, 2
    int S;       // complex systems per cell
    int nrows;   // rows of the system vector
    long nx, ny; // real storage extents of a row
    long ncx, ncy;  // cells per axis
    long ncells;
    long G;         // systems = ncells * S
    long mx_offset; // global x mode index of the first local cell (pencils sharded over ranks)
    const double *kx, *ky;
    int xtile;      // the STATE vector X -- the solution every solve writes, the input of every mat-vec -- (state_cell,
                    // state_row_stride; ddh_pencil_set_state_tiled): 0 natural [row][kx][ky]; 1 rows tile-major
                    // [row][kx/8][ky/8][8][8] like the right-hand-side vectors; 2 kx-band-major [kx/8][row][ky/8][8][8]
                    // (rows of a pencil 8 ny doubles apart instead of one nx ny plane)
};

struct MatDev {
    int nrows_out;
    int nterms;
    const int *rowptr;   // [nrows_out+1]  (terms sorted by row)
    const int *col;
    const double2 *coef;
    const unsigned *expo;  // ex | ey<<8 | dx<<16 | dy<<24
    const int *order;      // optional processing order of the rows (locality of the column accesses), or null
    // window form (band_matvec_kernel), when every term is a real, wavenumber-independent coefficient and the columns of a
    // row lie in a window of MV_W consecutive rows of x: band[r * MV_W + d] multiplies x[base[r] + d]; nt[r] = terms of
    // the row (0: the row of y is zero); null when the matrix does not have that form
    const double *band;
    const int *base;
};
constexpr int MV_W = 5;

// optional upper-banded back-substitution along the coupled index fused into a mat-vec:
// rows are (component, kz); y[comp, kz] = (A x)[comp, kz] - sum_{d>=1} band[d][kz] y[comp, kz+off_d]) / band[0][kz]
struct PostSolve {
    int nz;          // 0 -> disabled
    int nbands;
    int off[4];
    const double *bands;   // [nbands][nz]
};

struct Matrix {
    MatDev dev;
    std::vector<int> row_h, col_h;
    std::vector<double2> coef_h;
    std::vector<unsigned> expo_h;
    void *d_rowptr = nullptr, *d_col = nullptr, *d_coef = nullptr, *d_expo = nullptr, *d_order = nullptr;
    void *d_band = nullptr, *d_base = nullptr;
};

struct LuDev {
    int n, nb, N, kl, ku, W, BW;   // W = ku + kl; BW = entries stored per band row: kl + W + 1 (complex factors), real
                                   // factors: kpad + kl + (backward register window) + 1, rounded up to even
    int real;                      // 1: real graded matrix shared by the systems of a cell (see factor_real)
    long GL;                       // number of stored factorizations: G (complex) or ncells (real)
    long nblk;                     // ceil(GL / 64): factor storage is tiled [block of 64][row][entry][lane]
    int rows_aw;                   // rows per block: max(n, 1) (+ kl + kpad zero rows for real factors)
    int pk, pk63;                  // real factors: band entries stored in pairs per lane (lu_eoff); pk63 = pk ? 63 : 0
    int kpad;                      // real factors: zero entries in front of every band row, so that the forward sweep's
                                   // window of kl + kpad multipliers per column needs no guards (solve_forward_lean_kernel)
    void *Aw;                      // [block][rows_aw][BW][lane]  band rows (lu_aw), LAPACK-style fill space (double2 / double)
    void *Ab;                      // [N][nb][GL]  border rows (multipliers | Schur block inverse)
    unsigned char *piv;            // [n][GL]
    unsigned char *flag;           // [GL]
    double2 *scratch;              // [max(n, nb*nb)][G]
    const int *rowperm, *colperm;  // logical -> physical
    const unsigned char *row_axes, *col_axes;   // per logical border row / col: validity bits
    const unsigned char *row_code, *col_code;   // real mode, per logical row / col: bit0 rotate by i, bit1 sign for -kx
    // column recombination X = P Y fused into the backward sweep (ddh_pencil_solve_recombined): P in logical (graded)
    // ordering is unit upper banded, pband[j * PBW + d - 1] = P[j, j + d], d = 1..PBW; null = not fused
    const double *pband;
    // Partner pencils (ddh_pencil_set_pairing): when the problem is symmetric under the exchange of the two Fourier axes,
    // lambda(ky, kx) = Pi_r lambda(kx, ky) Pi_c, the cell (my, mx) is solved with the factorization of (mx, my): one
    // stored factorization ("slot") serves the P, Q systems of both cells, the four threads sit in adjacent lanes and
    // their factor loads coalesce -- the factor stream, 7/8 of the solve's bytes, halves.
    //   thread g -> virtual cell v = g / S: vcell[v] = the cell it solves, vslot[v] = 2 * slot + (1 for the partner member);
    //   the two members of a pair are consecutive virtual cells, the unpaired cells (axes, diagonal) follow with slots of
    //   their own -- every lane works, the thread count stays ncells * S (one resident round of wavefronts at 512^2)
    //   partner rows / columns are read and written through rowperm2 / colperm2 (= swap o perm); its Q system is the
    //   conjugate problem: lambda(kx, -ky) = conj lambda(-kx, ky)  (real operators), so the data are conjugated on the way
    //   in and out.
    // Independent diagonal blocks (ddh_pencil_set_row_blocks): the band block is block diagonal with nsplit blocks of nh
    // rows each (connected components of the pencil matrix, e.g. the two reflection parities of a problem between two
    // plates).  The one-thread-per-system sweeps of the real-graded path then run one thread per (system, block):
    // thread t -> block t / Gp, system t % Gp (Gp = G rounded up to the workgroup size); the dependent chain of a sweep is
    // nh rows instead of n.  nsplit = 1, nh = n otherwise.
    int nsplit, nh;
    long Gp;
    int pair;
    // Per-row fill of U, measured after every factorization (lu_width_kernel): wrow[j] = the last non-zero super-diagonal
    // of row j over ALL stored factorizations.  Partial pivoting may fill kl + ku super-diagonals, most rows use fewer
    // (512^2 x 256 Rayleigh-Benard: 15 of 17 for 486 of the 1288 rows, mean 13.5): the backward sweep loads only the
    // entry pairs that can hold data (-17 % of its factor bytes).  null = not measured (every pair is loaded).
    const int *wrow;
    const long *vcell;             // [ncells]
    const int *vslot;              // [ncells]
    const long *slot_cell;         // [GL] the cell whose matrix a slot holds
    const int *rowperm2, *colperm2;
};

constexpr int PBW = 12;

struct LuFactor {
    LuDev dev;
    size_t bytes = 0;
    void *d_rowperm = nullptr, *d_colperm = nullptr, *d_raxes = nullptr, *d_caxes = nullptr;
    void *d_rcode = nullptr, *d_ccode = nullptr;
    // dense fallback
    int nflag = 0;
    std::vector<long> flag_cells;   // flagged cell ids
    void *d_flag_cells = nullptr;   // long[nflagcells]
    void *d_inv = nullptr;          // double2 [nflagcells*S][N][N]
    void *d_dense_rhs = nullptr;    // double2 [nflagcells*S][N]
    std::vector<int> colperm_h;             // logical -> physical columns (host copy)
    std::vector<unsigned char> ccode_h;     // grading codes of the logical columns (real mode)
    void *d_rhs_tmp = nullptr;              // materialised right-hand side for the cooperative forward sweep
    // few systems, one Fourier axis: explicit inverses of the independent diagonal blocks, applied by blockinv_solve_kernel
    // instead of the sweeps (ddh_pencil_set_block_inverse; caller-owned memory)
    const double *d_binv = nullptr;
    void *d_pband = nullptr;                // [n][PBW] band of the recombination, see LuDev::pband
    void *d_vcell = nullptr, *d_vslot = nullptr, *d_rowperm2 = nullptr, *d_colperm2 = nullptr;   // partner pencils
    void *d_wrow = nullptr;                 // [n] int, see LuDev::wrow
    std::vector<long> slot_cells_h;         // [2 * GL]: the cell of a slot and its partner (-1 = none)
    void *d_slot_cell = nullptr;
    int pband_mat = -1;                     // matrix id it was built from
};

struct PencilPack : HandleBase {
    PencilDev dev;
    void *d_kx = nullptr, *d_ky = nullptr;
    std::vector<Matrix *> mats;
    std::vector<LuFactor *> lus;
    std::vector<PostSolve> posts;
    std::vector<void *> post_mem;
    // sweep variant of ddh_pencil_solve: mode 1 = by the number of systems, 0 = one thread per system, 2 = cooperative;
    // fwd / cb >= 0 override the forward kernel (0 / 1) and the backward lanes per system (0, 4, 16).  Initialised
    // once from DDH_SOLVE_COOP / DDH_COOP_FWD / DDH_COOP_CB when the pack is created (not per launch);
    // ddh_pencil_set_solve_variant changes them.
    int coop_mode = 1, coop_fwd = -1, coop_cb = -1;
    // x <-> y symmetry of the problem (ddh_pencil_set_pairing): physical row / column involutions, used by factorizations
    // of at least pair_min systems
    std::vector<int> pair_rows, pair_cols;
    long pair_min = 0;
    int row_blocks = 1;             // ddh_pencil_set_row_blocks: independent diagonal blocks of later factorizations
    std::vector<double> kx_h, ky_h;
    ~PencilPack() override;
};

static void free_lu(LuFactor *lu) {
    if (!lu) return;
    (void)hipFree(lu->dev.Aw);
    (void)hipFree(lu->dev.Ab);
    (void)hipFree(lu->dev.piv);
    (void)hipFree(lu->dev.flag);
    (void)hipFree(lu->dev.scratch);
    (void)hipFree(lu->d_rowperm);
    (void)hipFree(lu->d_colperm);
    (void)hipFree(lu->d_raxes);
    (void)hipFree(lu->d_caxes);
    (void)hipFree(lu->d_rcode);
    (void)hipFree(lu->d_ccode);
    (void)hipFree(lu->d_flag_cells);
    (void)hipFree(lu->d_inv);
    (void)hipFree(lu->d_dense_rhs);
    (void)hipFree(lu->d_pband);
    (void)hipFree(lu->d_rhs_tmp);
    (void)hipFree(lu->d_vcell);
    (void)hipFree(lu->d_vslot);
    (void)hipFree(lu->d_slot_cell);
    (void)hipFree(lu->d_rowperm2);
    (void)hipFree(lu->d_colperm2);
    (void)hipFree(lu->d_wrow);
    delete lu;
}

PencilPack::~PencilPack() {
    (void)hipFree(d_kx);
    (void)hipFree(d_ky);
    for (auto m : mats) {
        (void)hipFree(m->d_rowptr);
        (void)hipFree(m->d_col);
        (void)hipFree(m->d_coef);
        (void)hipFree(m->d_expo);
        if (m->d_order) (void)hipFree(m->d_order);
        if (m->d_band) (void)hipFree(m->d_band);
        if (m->d_base) (void)hipFree(m->d_base);
        delete m;
    }
    for (auto l : lus) free_lu(l);
    for (auto m : post_mem) (void)hipFree(m);
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ void cfma(double2 &acc, double2 a, double2 b) {   // acc += a*b
    acc.x += a.x * b.x - a.y * b.y;
    acc.y += a.x * b.y + a.y * b.x;
}
__device__ __forceinline__ void cfms(double2 &acc, double2 a, double2 b) {   // acc -= a*b
    acc.x -= a.x * b.x - a.y * b.y;
    acc.y -= a.x * b.y + a.y * b.x;
}
__device__ __forceinline__ double cabs2(double2 a) { return a.x * a.x + a.y * a.y; }
__device__ __forceinline__ double2 cinv(double2 a) {
    const double d = 1.0 / (a.x * a.x + a.y * a.y);
    return make_double2(a.x * d, -a.y * d);
}
__device__ __forceinline__ double ipow(double k, unsigned e) {
    double r = 1.0, b = k;
    if (e & 1u) r *= b;
    b *= b;
    if (e & 2u) r *= b;
    b *= b;
    if (e & 4u) r *= b;
    return r;
}

// element type of a stored factorization: complex (general) or real (graded, see ddh_pencil_factor_real)
template <bool REAL> struct El;
template <> struct El<false> {
    typedef double2 T;
    static __device__ __forceinline__ T zero() { return make_double2(0.0, 0.0); }
    static __device__ __forceinline__ T one() { return make_double2(1.0, 0.0); }
    static __device__ __forceinline__ T mul(T a, T b) { return cmul(a, b); }
    static __device__ __forceinline__ T inv(T a) { return cinv(a); }
    static __device__ __forceinline__ double abs2(T a) { return cabs2(a); }
    static __device__ __forceinline__ bool is_zero(T a) { return a.x == 0.0 && a.y == 0.0; }
    static __device__ __forceinline__ void fms(T &acc, T m, T v) { cfms(acc, m, v); }
    static __device__ __forceinline__ void add(T &acc, double2 v) { acc.x += v.x; acc.y += v.y; }
    // complex right-hand sides
    static __device__ __forceinline__ void fms2(double2 &acc, T m, double2 v) { cfms(acc, m, v); }
    static __device__ __forceinline__ void fma2(double2 &acc, T m, double2 v) { cfma(acc, m, v); }
    static __device__ __forceinline__ double2 mul2(double2 a, T m) { return cmul(a, m); }
    static __device__ __forceinline__ T scale(T a, double f) { return make_double2(a.x * f, a.y * f); }
};
template <> struct El<true> {
    typedef double T;
    static __device__ __forceinline__ T zero() { return 0.0; }
    static __device__ __forceinline__ T one() { return 1.0; }
    static __device__ __forceinline__ T mul(T a, T b) { return a * b; }
    static __device__ __forceinline__ T inv(T a) { return 1.0 / a; }
    static __device__ __forceinline__ double abs2(T a) { return a * a; }
    static __device__ __forceinline__ bool is_zero(T a) { return a == 0.0; }
    static __device__ __forceinline__ void fms(T &acc, T m, T v) { acc -= m * v; }
    static __device__ __forceinline__ void add(T &acc, double2 v) { acc += v.x; }
    static __device__ __forceinline__ void fms2(double2 &acc, T m, double2 v) { acc.x -= m * v.x; acc.y -= m * v.y; }
    static __device__ __forceinline__ void fma2(double2 &acc, T m, double2 v) { acc.x += m * v.x; acc.y += m * v.y; }
    static __device__ __forceinline__ double2 mul2(double2 a, T m) { return make_double2(a.x * m, a.y * m); }
    static __device__ __forceinline__ T scale(T a, double f) { return a * f; }
};

// Wave-uniform buffer addressing for the streams of the sweeps: resource = (address of the wave's FIRST lane, 4 GiB
// range), a lane's address = resource + 32-bit lane offset (one vector register) + a uniform 32-bit row offset (scalar
// register) + the instruction's immediate.  A 64-bit vector pointer per stream (plus one per out-of-range immediate) and
// its two-instruction carry chains per load group are gone -- registers are what keeps these kernels from one more
// resident wave per SIMD.  The first lane's address must be the smallest of the wave (factor slots and systems ascend
// with the lane) and the wave must span < 4 GiB.
typedef unsigned ddh_u4v __attribute__((ext_vector_type(4)));
typedef unsigned ddh_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wave_rsrc(const void *first_lane_ptr) {
    const unsigned long long a = (unsigned long long)first_lane_ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void *base = (void *)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0xffffffff, 0x00020000);
}
__device__ __forceinline__ unsigned wave_lane_off(const void *lane_ptr) {      // byte distance to the wave's first lane
    const unsigned long long a = (unsigned long long)lane_ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (unsigned)(a - (((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double2 bload16(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uni_off) {
    const ddh_u4v q = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, uni_off, 0);
    return make_double2(__hiloint2double((int)q.y, (int)q.x), __hiloint2double((int)q.w, (int)q.z));
}
__device__ __forceinline__ double bload8(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uni_off) {
    const ddh_u2v q = __builtin_amdgcn_raw_buffer_load_b64(r, lane_off, uni_off, 0);
    return __hiloint2double((int)q.y, (int)q.x);
}

// Factor storage index, [block of 64 factorizations][row][entry][lane]: everything a wave ever reads of one array is one
// contiguous stream, a row is one BW * 512 B chunk.
// (entry d of a band row: d = kl + (column - row); the row starts with kpad zero entries, see LuDev::kpad)
// pk (real factors): the entries of a row are stored in PAIRS per lane, [entry / 2][lane][2]: one 16-byte load brings two
// consecutive band entries of a system.  The sweeps are bound by the number of vector-memory requests in flight, not by
// bytes (a wave can have 63 outstanding, a row of the backward sweep is 35 entries): half the requests per row.
// (branch-free: pair e / 2 starts at 128 (e / 2), member e & 1 follows it:  64 e - 63 (e & 1);  pk63 = 63 or 0)
__device__ __forceinline__ long lu_eoff(const LuDev &L, int e) {          // raw entry e of a row, relative to raw entry 0
    return ((long)e << 6) - (long)(L.pk63 * (e & 1));
}
__device__ __forceinline__ long lu_aw(const LuDev &L, long gl, int row, int d) {
    return ((((gl >> 6) * (long)L.rows_aw + row) * L.BW) << 6) + lu_eoff(L, d + L.kpad) + ((gl & 63) << (L.pk ? 1 : 0));
}
__device__ __forceinline__ long lu_ab(const LuDev &L, long gl, int col, int rb) {
    return ((((gl >> 6) * (long)L.N + col) * L.nb + rb) << 6) + (gl & 63);
}
__device__ __forceinline__ long lu_pv(const LuDev &L, long gl, int row) {
    return (((gl >> 6) * (long)L.rows_aw + row) << 6) + (gl & 63);
}

struct CellCtx {
    long mx, my;   // local cell indices (addressing)
    long gmx;      // global x mode index (k = 0 tests)
    double kx, ky;
};

__device__ __forceinline__ CellCtx cell_ctx(const PencilDev &P, long cell) {
    CellCtx c;
    if (P.nf == 2) {
        c.mx = cell / P.ncy;
        c.my = cell % P.ncy;
    } else {
        c.mx = cell;
        c.my = 0;
    }
    c.gmx = c.mx + P.mx_offset;
    c.kx = (P.nf >= 1) ? P.kx[c.mx] : 0.0;
    c.ky = (P.nf == 2) ? P.ky[c.my] : 0.0;
    return c;
}

// thread of a one-thread-per-system sweep -> the system it solves (see LuDev::pair)
struct SysId {
    long cell, gl, G;   // cell (addressing), stored factorization, stride of the scratch vectors
    int s;
    bool partner, ok;
};
template <bool REAL>
__device__ __forceinline__ SysId sys_id(const PencilDev &P, const LuDev &L, long g) {
    SysId id;
    id.partner = false;
    id.G = P.G;
    id.ok = g < P.G;                // G is a multiple of S, pairs never straddle the guard
    id.cell = id.ok ? g / P.S : 0;
    id.s = (int)(g % P.S);
    if (L.pair) {
        const int vs = L.vslot[id.cell];
        id.gl = vs >> 1;
        id.partner = (vs & 1) != 0;
        id.cell = L.vcell[id.cell];
    } else {
        id.gl = REAL ? id.cell : g;
    }
    return id;
}

// wave-uniform: every lane of the wave works and the four lanes of each quad share one stored factorization -- the waves
// solve_backward_ring_kernel takes; the others (unpaired cells of the axes and the diagonal, a ragged last wave) are left
// to solve_backward_kernel<..., DBG = 64>
__device__ __forceinline__ bool ring_wave(const SysId &id, int lane) {
    const long gl_q = __shfl(id.gl, lane & ~3);
    return __builtin_amdgcn_readfirstlane((int)(__ballot(id.ok && gl_q == id.gl) == ~0ull)) != 0;
}

// real factor of a term for the +kx system; the -kx system multiplies by (-1)^ex
__device__ __forceinline__ double term_factor(unsigned e, const CellCtx &c) {
    const unsigned ex = e & 0xffu, ey = (e >> 8) & 0xffu, dx = (e >> 16) & 0xffu, dy = (e >> 24) & 0xffu;
    double f = ipow(c.kx, ex) * ipow(c.ky, ey);
    if (dx && c.gmx != 0) f = 0.0;
    if (dy && c.my != 0) f = 0.0;
    return f;
}

// validity of a border row/col for a cell: bit0 set -> exists for mx>0, bit1 set -> exists for my>0
__device__ __forceinline__ bool axes_valid(unsigned char bits, const CellCtx &c, int nf) {
    if (nf >= 1 && c.gmx != 0 && !(bits & 1)) return false;
    if (nf == 2 && c.my != 0 && !(bits & 2)) return false;
    return true;
}

// Tile-major layout of a row [nx][ny] of a system vector (two Fourier axes; nx, ny multiples of 8): the 64-byte segments
// (one storage row kx, four cells = 8 doubles along ky) of a tile of 4 x 4 cells -- 8 storage rows x 1 segment -- are
// contiguous: [kx / 8][ky / 8][kx % 8][ky % 8].  The sweeps give a wavefront the cells of one tile and of its transposed
// partner tile: in this layout its 64 16-byte accesses to a row are two contiguous 512-byte runs instead of sixteen
// 64-byte runs spread over eight storage rows (natural layout [kx][ky]).  Used for the solver-internal right-hand-side
// vectors (M.X, F); the state X keeps the natural layout its other consumers read.
__device__ __forceinline__ long tile_offset(long kxrow, long ky, long ny) {
    return (((kxrow >> 3) * (ny >> 3) + (ky >> 3)) << 6) + ((kxrow & 7) << 3) + (ky & 7);
}

// Element (row, storage row kxrow, ky) of a STATE vector (PencilDev::xtile) lies at row * state_row_stride + state_cell.
__device__ __forceinline__ long state_row_stride(const PencilDev &P) {
    return P.xtile == 2 ? 8 * P.ny : P.nx * P.ny;
}
__device__ __forceinline__ long state_cell(const PencilDev &P, long kxrow, long ky) {
    if (P.xtile == 2) return (kxrow >> 3) * ((long)P.nrows * 8 * P.ny) + ((ky >> 3) << 6) + ((kxrow & 7) << 3) + (ky & 7);
    return P.xtile ? tile_offset(kxrow, ky, P.ny) : kxrow * P.ny + ky;
}

// ------------------------------------------------------------------------------------------------
// y = A x for every cell (apply_sparse over all pencils; M.X and L.X of timesteppers.py:588-604)
// ------------------------------------------------------------------------------------------------
template <int NF>
__global__ void __launch_bounds__(256)
matvec_kernel(PencilDev P, MatDev A, const double *__restrict__ x, double *y, PostSolve ps, int rows_per_chunk) {
    const long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= P.ncells) return;
    const CellCtx c = cell_ctx(P, cell);
    const long plane = P.nx * P.ny;
    long off0, off1 = 0;
    long xoff0, xoff1 = 0;           // x: natural or tile-major (PencilDev::xtile); y: natural
    if (NF == 2) {
        off0 = (2 * c.mx) * P.ny + 2 * c.my;
        off1 = off0 + P.ny;
        xoff0 = state_cell(P, 2 * c.mx, 2 * c.my);
        xoff1 = state_cell(P, 2 * c.mx + 1, 2 * c.my);
    } else if (NF == 1) {
        off0 = xoff0 = 2 * c.mx;
    } else {
        off0 = xoff0 = 0;
    }
    double2 h0[4], h1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h0[j] = h1[j] = make_double2(0.0, 0.0);
    // blockIdx.y splits the rows into independent chunks (one component per chunk with a post-solve):
    // a single sweep over all rows per thread would leave one wave per SIMD, i.e. latency bound
    const int rr0 = blockIdx.y * rows_per_chunk;
    const int rr1 = (rr0 + rows_per_chunk < A.nrows_out) ? rr0 + rows_per_chunk : A.nrows_out;
    for (int rr = rr0; rr < rr1; ++rr) {
        int r = A.order ? A.order[rr] : rr, kz = 0, comp0 = 0;
        if (ps.nz > 0) {
            // descending coupled index inside each component so that y[kz + off] is already final
            const int comp = rr / ps.nz;
            kz = ps.nz - 1 - (rr - comp * ps.nz);
            comp0 = comp * ps.nz;
            r = comp0 + kz;
        }
        double2 accP = make_double2(0.0, 0.0), accQ = accP;
        const int t1 = A.rowptr[r + 1];
        for (int t = A.rowptr[r]; t < t1; ++t) {
            const unsigned e = A.expo[t];
            const double f = term_factor(e, c);
            const double2 cf = A.coef[t];
            const double2 v = make_double2(cf.x * f, cf.y * f);
            const double *xr = x + (long)A.col[t] * (NF == 2 ? state_row_stride(P) : plane);
            if (NF == 2) {
                const double2 a = *reinterpret_cast<const double2 *>(xr + xoff0);   // cc, cs
                const double2 b = *reinterpret_cast<const double2 *>(xr + xoff1);   // sc, ss
                const double2 xP = make_double2(a.x - b.y, a.y + b.x);
                const double2 xQ = make_double2(a.x + b.y, a.y - b.x);
                cfma(accP, v, xP);
                const double sg = (e & 1u) ? -1.0 : 1.0;
                cfma(accQ, make_double2(v.x * sg, v.y * sg), xQ);
            } else if (NF == 1) {
                const double2 xP = *reinterpret_cast<const double2 *>(xr + xoff0);
                cfma(accP, v, xP);
            } else {
                accP.x += v.x * xr[0];
            }
        }
        double *yr = y + (long)r * plane;
        // stored real parts: v0 = (cc, cs) [or (a, b)], v1 = (sc, ss)
        double2 v0, v1 = make_double2(0.0, 0.0);
        if (NF == 2) {
            v0 = make_double2(0.5 * (accP.x + accQ.x), 0.5 * (accP.y + accQ.y));
            v1 = make_double2(0.5 * (accP.y - accQ.y), 0.5 * (accQ.x - accP.x));
        } else {
            v0 = accP;
        }
        if (ps.nz > 0) {
            // the conversion is real and wavenumber-independent: it acts on each real part alike.
            // h0/h1[j] hold the finished outputs at kz+1+j (register window, reset per component).
            if (kz == ps.nz - 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) h0[j] = h1[j] = make_double2(0.0, 0.0);
            }
            for (int d = 1; d < ps.nbands; ++d) {
                const int o = ps.off[d];
                if (kz + o < ps.nz) {
                    const double bnd = ps.bands[d * ps.nz + kz];
                    double2 p0 = make_double2(0.0, 0.0), p1 = p0;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (o - 1 == j) { p0 = h0[j]; p1 = h1[j]; }
                    v0.x -= bnd * p0.x; v0.y -= bnd * p0.y;
                    v1.x -= bnd * p1.x; v1.y -= bnd * p1.y;
                }
            }
            const double inv = 1.0 / ps.bands[kz];
            v0.x *= inv; v0.y *= inv; v1.x *= inv; v1.y *= inv;
#pragma unroll
            for (int j = 3; j > 0; --j) { h0[j] = h0[j - 1]; h1[j] = h1[j - 1]; }
            h0[0] = v0;
            h1[0] = v1;
        }
        if (NF == 2) {
            *reinterpret_cast<double2 *>(yr + off0) = v0;
            *reinterpret_cast<double2 *>(yr + off1) = v1;
        } else if (NF == 1) {
            *reinterpret_cast<double2 *>(yr + off0) = v0;
        } else {
            yr[0] = v0.x;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// y = A x for matrices in window form (MatDev::band): the mass matrix M of an IVP -- conversions between polynomial
// bases, a few real coefficients per row, columns within MV_W consecutive rows of the same field.  matvec_kernel reads
// every x row once per term that uses it (M.X: 6.3 GB read for 2.2 GB of x rows with terms, the re-reads are 2-4 row
// steps apart and miss the caches); here a thread keeps the MV_W rows around the diagonal in registers and loads each x
// row once per row chunk.  The arithmetic per term and the order of the terms are those of matvec_kernel (a zero
// coefficient adds an exact zero), so the two kernels agree bit for bit.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
band_matvec_kernel(PencilDev P, MatDev A, const double *__restrict__ x, double *y, int rows_per_chunk, int keep_empty,
                   int out_tiled) {
    long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= P.ncells) return;
    if (P.xtile && out_tiled) {
        // x and y both tile-major: threads follow the tiles (16 consecutive threads = the 4 x 4 cells of one 512-byte tile, a
        // wavefront = four neighbouring tiles: every row access of a wave is one 2 KiB run).  Which thread computes a cell
        // does not change its value.
        const long tile = cell >> 4, tpr = P.ncy >> 2;
        cell = ((tile / tpr) * 4 + ((cell >> 2) & 3)) * P.ncy + (tile % tpr) * 4 + (cell & 3);
    }
    const CellCtx c = cell_ctx(P, cell);
    const long plane = P.nx * P.ny;
    const long off0 = state_cell(P, 2 * c.mx, 2 * c.my), off1 = state_cell(P, 2 * c.mx + 1, 2 * c.my);
    const long xrs = state_row_stride(P);
    // (y tile-major: ddh_pencil_matvec_update_tiled)
    const long yoff0 = out_tiled ? tile_offset(2 * c.mx, 2 * c.my, P.ny) : (2 * c.mx) * P.ny + 2 * c.my;
    const long yoff1 = out_tiled ? tile_offset(2 * c.mx + 1, 2 * c.my, P.ny) : (2 * c.mx + 1) * P.ny + 2 * c.my;
    const int rr0 = blockIdx.y * rows_per_chunk;
    const int rr1 = (rr0 + rows_per_chunk < A.nrows_out) ? rr0 + rows_per_chunk : A.nrows_out;
    double2 wa[MV_W], wb[MV_W];                  // x rows wbase .. wbase + MV_W - 1: (cc, cs) and (sc, ss)
#pragma unroll
    for (int d = 0; d < MV_W; ++d) wa[d] = wb[d] = make_double2(0.0, 0.0);
    int wbase = -(1 << 30);
    const int ncol = P.nrows;
    auto load_row = [&](int col, double2 &a, double2 &b) {
        const double *xr = x + (long)(col < ncol ? col : ncol - 1) * xrs;        // (slots beyond the last row: zero coefficient)
        a = *reinterpret_cast<const double2 *>(xr + off0);
        b = *reinterpret_cast<const double2 *>(xr + off1);
    };
    for (int r = rr0; r < rr1; ++r) {
        const int base = A.base[r];              // wave-uniform; < 0: the row has no terms
        double2 v0 = make_double2(0.0, 0.0), v1 = v0;
        if (base >= 0) {
            int adv = base - wbase;
            if (adv < 0 || adv > MV_W) {         // first row of the chunk / a jump to another field: fill the window
#pragma unroll
                for (int d = 0; d < MV_W; ++d) load_row(base + d, wa[d], wb[d]);
            } else {
                while (adv-- > 0) {              // one new row per step (the usual case: adv == 1)
#pragma unroll
                    for (int d = 0; d + 1 < MV_W; ++d) { wa[d] = wa[d + 1]; wb[d] = wb[d + 1]; }
                    ++wbase;
                    load_row(wbase + MV_W - 1, wa[MV_W - 1], wb[MV_W - 1]);
                }
            }
            wbase = base;
            const double *cf = A.band + (long)r * MV_W;
            double2 accP = make_double2(0.0, 0.0), accQ = accP;
#pragma unroll
            for (int d = 0; d < MV_W; ++d) {
                // (wave-uniform) window slots without a term are SKIPPED, not multiplied by zero: 0 * NaN / 0 * Inf of an
                // unrelated row of x must not reach y -- the term-list kernel never reads those rows
                if (cf[d] == 0.0) continue;
                const double2 v = make_double2(cf[d], 0.0);
                const double2 xP = make_double2(wa[d].x - wb[d].y, wa[d].y + wb[d].x);
                const double2 xQ = make_double2(wa[d].x + wb[d].y, wa[d].y - wb[d].x);
                cfma(accP, v, xP);
                cfma(accQ, v, xQ);
            }
            v0 = make_double2(0.5 * (accP.x + accQ.x), 0.5 * (accP.y + accQ.y));
            v1 = make_double2(0.5 * (accP.y - accQ.y), 0.5 * (accQ.x - accP.x));
        }
        if (base < 0 && keep_empty) continue;    // ddh_pencil_matvec_update: rows without terms are left as they are
        double *yr = y + (long)r * plane;
        *reinterpret_cast<double2 *>(yr + yoff0) = v0;
        *reinterpret_cast<double2 *>(yr + yoff1) = v1;
    }
}

// ------------------------------------------------------------------------------------------------
// The banded back-substitution of a mat-vec as its own pass (few cells): y[comp, kz] <- (y[comp, kz] -
// sum_{d>=1} band[d][kz] y[comp, kz+off_d]) / band[0][kz], kz descending.  Fused into the mat-vec (large problems:
// no second pass over y) the recurrence serializes the whole row computation behind it -- one memory latency per kz;
// for a few hundred cells it is cheaper to form all rows in parallel first and run only this light recurrence
// sequentially, with the next rows' values prefetched (their addresses do not depend on the recurrence).
// ------------------------------------------------------------------------------------------------
template <int NF>
__global__ void __launch_bounds__(64)
postsolve_kernel(PencilDev P, double *y, PostSolve ps) {
    const long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= P.ncells) return;
    const CellCtx c = cell_ctx(P, cell);
    const long plane = P.nx * P.ny;
    long off0, off1 = 0;
    if (NF == 2) {
        off0 = (2 * c.mx) * P.ny + 2 * c.my;
        off1 = off0 + P.ny;
    } else if (NF == 1) {
        off0 = 2 * c.mx;
    } else {
        off0 = 0;
    }
    const int comp0 = blockIdx.y * ps.nz;
    double2 h0[4], h1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h0[j] = h1[j] = make_double2(0.0, 0.0);
    constexpr int PF = 8;
    for (int k0 = ps.nz - 1; k0 >= 0; k0 -= PF) {
        // everything the next PF rows need -- their values AND their band coefficients -- is requested before the
        // recurrence touches the first of them
        double2 a0[PF], a1[PF];
        double bnd[PF][4];
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int kz = k0 - q;
            a0[q] = a1[q] = make_double2(0.0, 0.0);
#pragma unroll
            for (int d = 0; d < 4; ++d) bnd[q][d] = 0.0;
            if (kz >= 0) {
                const double *yr = y + (long)(comp0 + kz) * plane;
                if (NF == 0) a0[q].x = yr[0];
                else a0[q] = *reinterpret_cast<const double2 *>(yr + off0);
                if (NF == 2) a1[q] = *reinterpret_cast<const double2 *>(yr + off1);
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    if (d < ps.nbands) bnd[q][d] = ps.bands[d * ps.nz + kz];
            }
        }
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int kz = k0 - q;
            if (kz < 0) break;
            double2 v0 = a0[q], v1 = a1[q];
#pragma unroll
            for (int d = 1; d < 4; ++d) {
                if (d < ps.nbands) {
                    const int o = ps.off[d];
                    if (kz + o < ps.nz) {
                        const double b = bnd[q][d];
                        double2 p0 = make_double2(0.0, 0.0), p1 = p0;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (o - 1 == j) { p0 = h0[j]; p1 = h1[j]; }
                        v0.x -= b * p0.x; v0.y -= b * p0.y;
                        v1.x -= b * p1.x; v1.y -= b * p1.y;
                    }
                }
            }
            const double inv = 1.0 / bnd[q][0];
            v0.x *= inv; v0.y *= inv; v1.x *= inv; v1.y *= inv;
#pragma unroll
            for (int j = 3; j > 0; --j) { h0[j] = h0[j - 1]; h1[j] = h1[j - 1]; }
            h0[0] = v0;
            h1[0] = v1;
            double *yr = y + (long)(comp0 + kz) * plane;
            if (NF == 2) {
                *reinterpret_cast<double2 *>(yr + off0) = v0;
                *reinterpret_cast<double2 *>(yr + off1) = v1;
            } else if (NF == 1) {
                *reinterpret_cast<double2 *>(yr + off0) = v0;
            } else {
                yr[0] = v0.x;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// system <-> real storage (used by factor's nothing, solve's RHS load and solution store)
// thread g = cell*S + s.  For NF == 2 lanes (2i, 2i+1) hold the P and Q systems of one cell.
// ------------------------------------------------------------------------------------------------
template <int NF, int XD = 1>
__device__ __forceinline__ double2 load_sys(const double *__restrict__ v, long plane, int row, const PencilDev &P,
                                            const CellCtx &c, int s) {
    const double *vr = v + (long)row * plane;
    if (NF == 2) {
        const double2 mine = *reinterpret_cast<const double2 *>(vr + (2 * c.mx + s) * P.ny + 2 * c.my);
        double2 other;
        other.x = __shfl_xor(mine.x, XD);
        other.y = __shfl_xor(mine.y, XD);
        // s==0: mine=(cc,cs), other=(sc,ss) -> P ; s==1: mine=(sc,ss), other=(cc,cs) -> Q
        return s == 0 ? make_double2(mine.x - other.y, mine.y + other.x)
                      : make_double2(other.x + mine.y, other.y - mine.x);
    } else if (NF == 1) {
        return *reinterpret_cast<const double2 *>(vr + 2 * c.mx);
    } else {
        return make_double2(vr[0], 0.0);
    }
}

// Right-hand side of a solve given as a linear combination of stored system vectors (RHS assembly of
// timesteppers.py:156-166, 617-623 fused into the forward sweep: the combined vector is never written to HBM).
constexpr int RHS_MAX = 8;
struct RhsSrc {
    int n;
    const double *p[RHS_MAX];
    double a[RHS_MAX];
    const unsigned char *zrow;   // optional, per row of the system vectors: 1 = the row is zero in EVERY term (not read)
    const unsigned char *skip;   // optional, per row of the solution: 1 = the caller does not need the row (not written)
    int tiled;                   // 1: the term vectors are stored tile-major (tile_offset), see ddh_pencil_solve_recombined_tiled
#ifdef DDH_SWEEP_ABLATE
    int abl;                     // timing ablations (results are NOT a solve): DDH_ABL bit mask, see launch_solve
#endif
};

template <int NF, int XD = 1>
__device__ __forceinline__ double2 load_sys(const RhsSrc &r, long plane, int row, const PencilDev &P,
                                            const CellCtx &c, int s) {
    // (compile-time indices into the by-value argument struct: a run-time index would push it to scratch memory)
    if (r.n == 1 && !r.tiled) {
        double2 v = load_sys<NF, XD>(r.p[0], plane, row, P, c, s);
        if (r.a[0] != 1.0) v = make_double2(r.a[0] * v.x, r.a[0] * v.y);
        return v;
    }
    const long roff = (long)row * plane;
    if (NF == 2) {
        // (r.tiled: tile-major term vectors, see tile_offset; uniform, the offset is loop-invariant per thread)
        const long off = roff + (r.tiled ? tile_offset(2 * c.mx + s, 2 * c.my, P.ny) : (2 * c.mx + s) * P.ny + 2 * c.my);
        double2 mine = make_double2(0.0, 0.0);
#pragma unroll
        for (int t = 0; t < RHS_MAX; ++t) {
            if (t < r.n) {
                const double2 v = *reinterpret_cast<const double2 *>(r.p[t] + off);
                mine.x += r.a[t] * v.x;
                mine.y += r.a[t] * v.y;
            }
        }
        double2 other;
        other.x = __shfl_xor(mine.x, XD);
        other.y = __shfl_xor(mine.y, XD);
        return s == 0 ? make_double2(mine.x - other.y, mine.y + other.x)
                      : make_double2(other.x + mine.y, other.y - mine.x);
    } else if (NF == 1) {
        double2 acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int t = 0; t < RHS_MAX; ++t) {
            if (t < r.n) {
                const double2 v = *reinterpret_cast<const double2 *>(r.p[t] + roff + 2 * c.mx);
                acc.x += r.a[t] * v.x;
                acc.y += r.a[t] * v.y;
            }
        }
        return acc;
    } else {
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < RHS_MAX; ++t)
            if (t < r.n) acc += r.a[t] * r.p[t][roff];
        return make_double2(acc, 0.0);
    }
}

template <int NF, int XD = 1>
__device__ __forceinline__ void store_sys(double *__restrict__ v, long plane, int row, const PencilDev &P,
                                          const CellCtx &c, int s, double2 val, bool writer = true) {
    double *vr = v + (long)row * (NF == 2 ? state_row_stride(P) : plane);
    if (NF == 2) {
        double2 other;
        other.x = __shfl_xor(val.x, XD);
        other.y = __shfl_xor(val.y, XD);
        double2 out = (s == 0) ? make_double2(0.5 * (val.x + other.x), 0.5 * (val.y + other.y))    // cc, cs
                               : make_double2(0.5 * (other.y - val.y), 0.5 * (val.x - other.x));  // sc, ss
        // the msin parts of a k = 0 mode are not modes of a real field (the reference never stores them: valid-mode
        // filtering, core/subsystems.py:540-556).  The P and Q systems of such a cell are complex conjugates solved
        // independently, which would leave round-off there: write exact zeros.
        if (c.my == 0) out.y = 0.0;
        if (c.gmx == 0 && s == 1) out = make_double2(0.0, 0.0);
        if (writer) *reinterpret_cast<double2 *>(vr + state_cell(P, 2 * c.mx + s, 2 * c.my)) = out;
    } else if (NF == 1) {
        if (c.gmx == 0) val.y = 0.0;
        if (writer) *reinterpret_cast<double2 *>(vr + 2 * c.mx) = val;
    } else {
        if (writer) vr[0] = val.x;
    }
}

// ------------------------------------------------------------------------------------------------
// assemble a*M + b*L into band + border storage and factor it, one thread per system
// ------------------------------------------------------------------------------------------------
template <bool REAL>
__device__ __forceinline__ void scatter_terms(const PencilDev &P, const LuDev &L, const MatDev &A, double scale,
                                              const int *__restrict__ rowinv, const int *__restrict__ colinv,
                                              const CellCtx &c, int s, long g, double &anorm, bool &bad) {
    typedef typename El<REAL>::T E;
    E *Aw = (E *)L.Aw, *Ab = (E *)L.Ab;
    const long GL = L.GL;
    if (scale == 0.0) return;
    for (int r = 0; r < A.nrows_out; ++r) {
        const int i = rowinv[r];
        const int t1 = A.rowptr[r + 1];
        for (int t = A.rowptr[r]; t < t1; ++t) {
            const unsigned e = A.expo[t];
            double f = term_factor(e, c) * scale;
            if (s == 1 && (e & 1u)) f = -f;
            if (f == 0.0) continue;
            const double2 cf = A.coef[t];
            const double2 v = make_double2(cf.x * f, cf.y * f);
            const int cc = colinv[A.col[t]];
            anorm = fmax(anorm, fabs(v.x) + fabs(v.y));
            if (i < L.n) {
                const int d = cc - i + L.kl;
                if (d < 0 || cc - i > L.ku) {
                    // entry outside the band (e.g. a k=0-only gauge column): this pencil goes to
                    // the dense fallback
                    bad = true;
                    continue;
                }
                E *p = Aw + lu_aw(L, g, i, d);
                E o = *p;
                El<REAL>::add(o, v);
                *p = o;
            } else {
                E *p = Ab + lu_ab(L, g, cc, i - L.n);
                E o = *p;
                El<REAL>::add(o, v);
                *p = o;
            }
        }
    }
}

template <bool REAL>
__global__ void __launch_bounds__(64)
factor_kernel(PencilDev P, LuDev L, MatDev M, MatDev Lm, double a, double b, const int *__restrict__ rowinv,
              const int *__restrict__ colinv) {
    typedef typename El<REAL>::T E;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;   // index of the stored factorization
    if (g >= L.GL) return;
    const long cell = REAL ? (L.pair ? L.slot_cell[g] : g) : g / P.S;
    const int s = REAL ? 0 : (int)(g % P.S);
    const CellCtx c = cell_ctx(P, cell);
    const long G = L.GL;
    E *Aw = (E *)L.Aw, *Ab = (E *)L.Ab;
    const int n = L.n, nb = L.nb, N = L.N, kl = L.kl, W = L.W, BW = L.BW;
    double anorm = 0.0;
    bool bad = false;
    scatter_terms<REAL>(P, L, M, a, rowinv, colinv, c, s, g, anorm, bad);
    scatter_terms<REAL>(P, L, Lm, b, rowinv, colinv, c, s, g, anorm, bad);
    // border rows / columns that do not exist for this cell are paired into identity entries
    {
        int cb = 0;
        for (int rb = 0; rb < nb; ++rb) {
            if (axes_valid(L.row_axes[rb], c, P.nf)) continue;
            while (cb < nb && axes_valid(L.col_axes[cb], c, P.nf)) ++cb;
            if (cb < nb) {
                Ab[lu_ab(L, g, n + cb, rb)] = El<REAL>::one();
                ++cb;
            }
        }
    }
    const double tiny = 1e-13 * anorm;
    bad = bad || (anorm == 0.0);
    // ---- band elimination with partial pivoting among rows j..j+kl (border rows never pivot)
    for (int j = 0; j < n; ++j) {
        int p = 0;
        double best = -1.0;
        const int imax = (j + kl < n) ? kl : (n - 1 - j);
        for (int i = 0; i <= imax; ++i) {
            const double m = El<REAL>::abs2(Aw[lu_aw(L, g, j + i, kl - i)]);
            if (m > best) {
                best = m;
                p = i;
            }
        }
        L.piv[lu_pv(L, g, j)] = (unsigned char)p;
        const int wmax = (j + W < N) ? W : (N - 1 - j);   // columns j .. j+wmax
        if (p != 0) {
            for (int d = 0; d <= wmax; ++d) {
                E *pa = Aw + lu_aw(L, g, j, kl + d);
                E *pb = Aw + lu_aw(L, g, j + p, (kl - p) + d);
                const E t = *pa;
                *pa = *pb;
                *pb = t;
            }
        }
        E piv = Aw[lu_aw(L, g, j, kl)];
        if (!(El<REAL>::abs2(piv) > tiny * tiny)) {
            bad = true;
            piv = El<REAL>::one();
            Aw[lu_aw(L, g, j, kl)] = piv;
        }
        const E ip = El<REAL>::inv(piv);
        for (int i = 1; i <= imax; ++i) {
            E *pm = Aw + lu_aw(L, g, j + i, kl - i);
            const E m = El<REAL>::mul(*pm, ip);
            *pm = m;
            if (El<REAL>::is_zero(m)) continue;
            for (int d = 1; d <= wmax; ++d) {
                E *pt = Aw + lu_aw(L, g, j + i, (kl - i) + d);
                E t = *pt;
                El<REAL>::fms(t, m, Aw[lu_aw(L, g, j, kl + d)]);
                *pt = t;
            }
        }
        for (int rb = 0; rb < nb; ++rb) {
            E *pm = Ab + lu_ab(L, g, j, rb);
            const E m = El<REAL>::mul(*pm, ip);
            *pm = m;
            if (El<REAL>::is_zero(m)) continue;
            for (int d = 1; d <= wmax; ++d) {
                E *pt = Ab + lu_ab(L, g, j + d, rb);
                E t = *pt;
                El<REAL>::fms(t, m, Aw[lu_aw(L, g, j, kl + d)]);
                *pt = t;
            }
        }
        // store the reciprocal pivot: the solve multiplies instead of dividing
        Aw[lu_aw(L, g, j, kl)] = ip;
    }
    // ---- Schur block (nb x nb) at Ab[n + c][r]: invert in place by Gauss-Jordan with pivoting.
    if (nb > 0) {
        E *Sinv = (E *)L.scratch;   // [nb*nb][GL] workspace (scratch holds >= nb*nb*G complex)
        for (int r = 0; r < nb; ++r)
            for (int cidx = 0; cidx < nb; ++cidx)
                Sinv[((long)r * nb + cidx) * G + g] = (r == cidx) ? El<REAL>::one() : El<REAL>::zero();
        for (int k = 0; k < nb; ++k) {
            int p = k;
            double best = -1.0;
            for (int r = k; r < nb; ++r) {
                const double m = El<REAL>::abs2(Ab[lu_ab(L, g, n + k, r)]);
                if (m > best) {
                    best = m;
                    p = r;
                }
            }
            if (p != k) {
                for (int cidx = 0; cidx < nb; ++cidx) {
                    E *pa = Ab + lu_ab(L, g, n + cidx, k), *pb = Ab + lu_ab(L, g, n + cidx, p);
                    E t = *pa;
                    *pa = *pb;
                    *pb = t;
                    pa = Sinv + ((long)k * nb + cidx) * G + g;
                    pb = Sinv + ((long)p * nb + cidx) * G + g;
                    t = *pa;
                    *pa = *pb;
                    *pb = t;
                }
            }
            E piv = Ab[lu_ab(L, g, n + k, k)];
            if (!(El<REAL>::abs2(piv) > tiny * tiny)) {
                bad = true;
                piv = El<REAL>::one();
            }
            const E ip = El<REAL>::inv(piv);
            for (int cidx = 0; cidx < nb; ++cidx) {
                E *pa = Ab + lu_ab(L, g, n + cidx, k);
                *pa = El<REAL>::mul(*pa, ip);
                pa = Sinv + ((long)k * nb + cidx) * G + g;
                *pa = El<REAL>::mul(*pa, ip);
            }
            for (int r = 0; r < nb; ++r) {
                if (r == k) continue;
                const E m = Ab[lu_ab(L, g, n + k, r)];
                if (El<REAL>::is_zero(m)) continue;
                for (int cidx = 0; cidx < nb; ++cidx) {
                    E *pt = Ab + lu_ab(L, g, n + cidx, r);
                    E t = *pt;
                    El<REAL>::fms(t, m, Ab[lu_ab(L, g, n + cidx, k)]);
                    *pt = t;
                    pt = Sinv + ((long)r * nb + cidx) * G + g;
                    t = *pt;
                    El<REAL>::fms(t, m, Sinv[((long)k * nb + cidx) * G + g]);
                    *pt = t;
                }
            }
        }
        // copy inverse into the Schur slot: Ab[n + c][r] = Sinv[r][c]
        for (int r = 0; r < nb; ++r)
            for (int cidx = 0; cidx < nb; ++cidx)
                Ab[lu_ab(L, g, n + cidx, r)] = Sinv[((long)r * nb + cidx) * G + g];
    }
    L.flag[g] = bad ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// The same factorization with the rows of an elimination step spread over the workgroup (round 3).
// factor_kernel gives a whole system to one thread: 33 151 systems are 518 wavefronts -- half a wave per SIMD -- each
// walking 1289 steps of ~450 dependent global read-modify-writes: 135 ms for 512 x 512 x 256, latency from end to end.
// Here a workgroup is 64 systems (threadIdx.x, the coalescing lane of the factor layout, unchanged) x NT = kl + 1 + nb
// row threads (threadIdx.y): thread (x, i) owns window row j + i of system x (i <= kl) or border row i - kl - 1.  A step
// is three phases -- pivot search, row interchange, the kl (+ nb) independent row updates -- with a workgroup barrier in
// between; the assembly of a M + b L is split over the same threads by physical row.  Same arithmetic in the same
// order per entry as factor_kernel: the factors are bit-identical.
// ------------------------------------------------------------------------------------------------
constexpr int FR_NTMAX = 16;

template <bool REAL>
__device__ __forceinline__ void scatter_row(const PencilDev &P, const LuDev &L, const MatDev &A, double scale, int r, int i,
                                            const int *__restrict__ colinv, const CellCtx &c, int s, long g, double &anorm,
                                            bool &bad) {
    typedef typename El<REAL>::T E;
    E *Aw = (E *)L.Aw, *Ab = (E *)L.Ab;
    if (scale == 0.0 || r >= A.nrows_out) return;
    const int t1 = A.rowptr[r + 1];
    for (int t = A.rowptr[r]; t < t1; ++t) {
        const unsigned e = A.expo[t];
        double f = term_factor(e, c) * scale;
        if (s == 1 && (e & 1u)) f = -f;
        if (f == 0.0) continue;
        const double2 cf = A.coef[t];
        const double2 v = make_double2(cf.x * f, cf.y * f);
        const int cc = colinv[A.col[t]];
        anorm = fmax(anorm, fabs(v.x) + fabs(v.y));
        if (i < L.n) {
            const int d = cc - i + L.kl;
            if (d < 0 || cc - i > L.ku) {
                bad = true;
                continue;
            }
            E *p = Aw + lu_aw(L, g, i, d);
            E o = *p;
            El<REAL>::add(o, v);
            *p = o;
        } else {
            E *p = Ab + lu_ab(L, g, cc, i - L.n);
            E o = *p;
            El<REAL>::add(o, v);
            *p = o;
        }
    }
}

template <bool REAL>
__global__ void __launch_bounds__(64 * FR_NTMAX)
factor_rows_kernel(PencilDev P, LuDev L, MatDev M, MatDev Lm, double a, double b, const int *__restrict__ rowinv,
                   const int *__restrict__ colinv) {
    typedef typename El<REAL>::T E;
    const int tx = threadIdx.x, ty = threadIdx.y, NT = blockDim.y;
    // the factor storage is allocated in whole blocks of 64: lanes beyond the last system factor an all-zero matrix in
    // their own (unused) slot, with the cell data of the last system
    const long g = (long)blockIdx.x * 64 + tx;
    const long gc = g < L.GL ? g : L.GL - 1;
    const long cell = REAL ? (L.pair ? L.slot_cell[gc] : gc) : gc / P.S;
    const int s = REAL ? 0 : (int)(gc % P.S);
    const CellCtx c = cell_ctx(P, cell);
    const long G = L.GL;
    E *Aw = (E *)L.Aw, *Ab = (E *)L.Ab;
    const int n = L.n, nb = L.nb, N = L.N, kl = L.kl, W = L.W;
    __shared__ double s_val[FR_NTMAX][64];
    __shared__ int s_flag[FR_NTMAX][64];
    double anorm = 0.0;
    bool bad = false;
    if (g < L.GL) {
        for (int r = ty; r < N; r += NT) {
            const int i = rowinv[r];
            scatter_row<REAL>(P, L, M, a, r, i, colinv, c, s, g, anorm, bad);
            scatter_row<REAL>(P, L, Lm, b, r, i, colinv, c, s, g, anorm, bad);
        }
    }
    s_val[ty][tx] = anorm;
    s_flag[ty][tx] = bad ? 1 : 0;
    __syncthreads();
    for (int t = 0; t < NT; ++t) {
        anorm = fmax(anorm, s_val[t][tx]);
        bad = bad || (s_flag[t][tx] != 0);
    }
    // border rows / columns that do not exist for this cell are paired into identity entries
    if (ty == 0) {
        int cb = 0;
        for (int rb = 0; rb < nb; ++rb) {
            if (axes_valid(L.row_axes[rb], c, P.nf)) continue;
            while (cb < nb && axes_valid(L.col_axes[cb], c, P.nf)) ++cb;
            if (cb < nb) {
                Ab[lu_ab(L, g, n + cb, rb)] = El<REAL>::one();
                ++cb;
            }
        }
    }
    const double tiny = 1e-13 * anorm;
    bad = bad || (anorm == 0.0);
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        const int imax = (j + kl < n) ? kl : (n - 1 - j);
        const int wmax = (j + W < N) ? W : (N - 1 - j);   // columns j .. j+wmax
        // ---- (A) pivot search among rows j .. j + imax
        s_val[ty < FR_NTMAX ? ty : 0][tx] = (ty <= imax) ? El<REAL>::abs2(Aw[lu_aw(L, g, j + ty, kl - ty)]) : -1.0;
        __syncthreads();
        int p = 0;
        {
            double best = -1.0;
            for (int i = 0; i <= imax; ++i) {
                const double m = s_val[i][tx];
                if (m > best) {
                    best = m;
                    p = i;
                }
            }
        }
        if (ty == 0) L.piv[lu_pv(L, g, j)] = (unsigned char)p;
        // ---- (B) row interchange j <-> j + p, the entries dealt out to the row threads
        if (p != 0) {
            for (int d = ty; d <= wmax; d += NT) {
                E *pa = Aw + lu_aw(L, g, j, kl + d);
                E *pb = Aw + lu_aw(L, g, j + p, (kl - p) + d);
                const E t = *pa;
                *pa = *pb;
                *pb = t;
            }
        }
        __syncthreads();
        // ---- (C) multipliers and row updates: one window row / border row per thread
        E piv = Aw[lu_aw(L, g, j, kl)];
        if (!(El<REAL>::abs2(piv) > tiny * tiny)) {
            bad = true;
            piv = El<REAL>::one();
        }
        const E ip = El<REAL>::inv(piv);
        if (ty >= 1 && ty <= imax) {
            E *pm = Aw + lu_aw(L, g, j + ty, kl - ty);
            const E m = El<REAL>::mul(*pm, ip);
            *pm = m;
            if (!El<REAL>::is_zero(m)) {
                for (int d = 1; d <= wmax; ++d) {
                    E *pt = Aw + lu_aw(L, g, j + ty, (kl - ty) + d);
                    E t = *pt;
                    El<REAL>::fms(t, m, Aw[lu_aw(L, g, j, kl + d)]);
                    *pt = t;
                }
            }
        } else if (ty > kl && ty - kl - 1 < nb) {
            const int rb = ty - kl - 1;
            E *pm = Ab + lu_ab(L, g, j, rb);
            const E m = El<REAL>::mul(*pm, ip);
            *pm = m;
            if (!El<REAL>::is_zero(m)) {
                for (int d = 1; d <= wmax; ++d) {
                    E *pt = Ab + lu_ab(L, g, j + d, rb);
                    E t = *pt;
                    El<REAL>::fms(t, m, Aw[lu_aw(L, g, j, kl + d)]);
                    *pt = t;
                }
            }
        }
        __syncthreads();
        // store the reciprocal pivot: the solve multiplies instead of dividing
        if (ty == 0) Aw[lu_aw(L, g, j, kl)] = ip;
    }
    __syncthreads();
    // ---- Schur block (nb x nb) at Ab[n + c][r]: invert in place by Gauss-Jordan with pivoting (one thread per system)
    if (ty == 0 && nb > 0 && g < L.GL) {
        E *Sinv = (E *)L.scratch;   // [nb*nb][GL] workspace
        const long gs = g;
        for (int r = 0; r < nb; ++r)
            for (int cidx = 0; cidx < nb; ++cidx)
                Sinv[((long)r * nb + cidx) * G + gs] = (r == cidx) ? El<REAL>::one() : El<REAL>::zero();
        for (int k = 0; k < nb; ++k) {
            int p = k;
            double best = -1.0;
            for (int r = k; r < nb; ++r) {
                const double m = El<REAL>::abs2(Ab[lu_ab(L, g, n + k, r)]);
                if (m > best) {
                    best = m;
                    p = r;
                }
            }
            if (p != k) {
                for (int cidx = 0; cidx < nb; ++cidx) {
                    E *pa = Ab + lu_ab(L, g, n + cidx, k), *pb = Ab + lu_ab(L, g, n + cidx, p);
                    E t = *pa;
                    *pa = *pb;
                    *pb = t;
                    pa = Sinv + ((long)k * nb + cidx) * G + gs;
                    pb = Sinv + ((long)p * nb + cidx) * G + gs;
                    t = *pa;
                    *pa = *pb;
                    *pb = t;
                }
            }
            E piv = Ab[lu_ab(L, g, n + k, k)];
            if (!(El<REAL>::abs2(piv) > tiny * tiny)) {
                bad = true;
                piv = El<REAL>::one();
            }
            const E ip = El<REAL>::inv(piv);
            for (int cidx = 0; cidx < nb; ++cidx) {
                E *pa = Ab + lu_ab(L, g, n + cidx, k);
                *pa = El<REAL>::mul(*pa, ip);
                pa = Sinv + ((long)k * nb + cidx) * G + gs;
                *pa = El<REAL>::mul(*pa, ip);
            }
            for (int r = 0; r < nb; ++r) {
                if (r == k) continue;
                const E m = Ab[lu_ab(L, g, n + k, r)];
                if (El<REAL>::is_zero(m)) continue;
                for (int cidx = 0; cidx < nb; ++cidx) {
                    E *pt = Ab + lu_ab(L, g, n + cidx, r);
                    E t = *pt;
                    El<REAL>::fms(t, m, Ab[lu_ab(L, g, n + cidx, k)]);
                    *pt = t;
                    pt = Sinv + ((long)r * nb + cidx) * G + gs;
                    t = *pt;
                    El<REAL>::fms(t, m, Sinv[((long)k * nb + cidx) * G + gs]);
                    *pt = t;
                }
            }
        }
        for (int r = 0; r < nb; ++r)
            for (int cidx = 0; cidx < nb; ++cidx)
                Ab[lu_ab(L, g, n + cidx, r)] = Sinv[((long)r * nb + cidx) * G + gs];
    }
    if (ty == 0 && g < L.GL) L.flag[g] = bad ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// Register-window variant of the row-parallel factorization: a row thread KEEPS its band row in registers for the kl + 1
// steps the row spends in the elimination window (row r belongs to thread r mod (kl + 1)).  A row is read from the
// assembled band storage once, when it enters the window, and written once, as a finished U row, when it becomes the
// pivot row; in between it is updated in registers against the pivot row, which the workgroup shares through LDS (as is
// the row displaced by an interchange).  factor_rows_kernel re-reads and re-writes the whole (kl + 1) x (W + 1) window
// from global memory every step: 7.5 MB per system against ~0.9 MB here.  Same operations per entry in the same order:
// bit-identical factors.
// ------------------------------------------------------------------------------------------------
constexpr int FW_WMAX = 36;      // window columns kl + ku + 1 held per thread (the general instantiation)
// The narrow instantiation (FW_WMAX_T_T = 20 columns, 8 row threads, at most 84 registers): THREE workgroups per CU.  The
// Rayleigh-Benard factorization at 512^2 is 514 workgroups of 8 waves; at 124 registers two of them fit a CU -- 512 slots
// for 514 workgroups, i.e. the last two ran alone after everyone else and the kernel took two rounds (16.9 ms, rounds 3-5).
constexpr int FW_WNARROW = 20, FW_NT_NARROW = 8;

template <bool REAL, int FW_WMAX_T = FW_WMAX, int NTB = FR_NTMAX, int MINB = 1>     // MINB: waves per SIMD the register budget is sized for
__global__ void __launch_bounds__(64 * NTB, MINB)
factor_window_kernel(PencilDev P, LuDev L, MatDev M, MatDev Lm, double a, double b, const int *__restrict__ rowinv,
                     const int *__restrict__ colinv) {
    typedef typename El<REAL>::T E;
    const int tx = threadIdx.x, ty = threadIdx.y, NT = blockDim.y;
    const long g = (long)blockIdx.x * 64 + tx;
    const long gc = g < L.GL ? g : L.GL - 1;
    const long cell = REAL ? (L.pair ? L.slot_cell[gc] : gc) : gc / P.S;
    const int s = REAL ? 0 : (int)(gc % P.S);
    const CellCtx c = cell_ctx(P, cell);
    const long G = L.GL;
    E *Aw = (E *)L.Aw, *Ab = (E *)L.Ab;
    const int n = L.n, nb = L.nb, N = L.N, kl = L.kl, W = L.W, KL1 = L.kl + 1;
    extern __shared__ double s_dyn_fw[];
    E *s_piv = (E *)s_dyn_fw;                         // [FW_WMAX_T][64] the pivot row of the step
    E *s_disp = s_piv + FW_WMAX_T * 64;                 // [FW_WMAX_T][64] the row an interchange displaces
    __shared__ double s_val[FR_NTMAX][64];
    __shared__ int s_flag[FR_NTMAX][64];
    double anorm = 0.0;
    bool bad = false;
    if (g < L.GL) {
        for (int r = ty; r < N; r += NT) {
            const int i = rowinv[r];
            scatter_row<REAL>(P, L, M, a, r, i, colinv, c, s, g, anorm, bad);
            scatter_row<REAL>(P, L, Lm, b, r, i, colinv, c, s, g, anorm, bad);
        }
    }
    s_val[ty][tx] = anorm;
    s_flag[ty][tx] = bad ? 1 : 0;
    __syncthreads();
    for (int t = 0; t < NT; ++t) {
        anorm = fmax(anorm, s_val[t][tx]);
        bad = bad || (s_flag[t][tx] != 0);
    }
    if (ty == 0) {
        int cb = 0;
        for (int rb = 0; rb < nb; ++rb) {
            if (axes_valid(L.row_axes[rb], c, P.nf)) continue;
            while (cb < nb && axes_valid(L.col_axes[cb], c, P.nf)) ++cb;
            if (cb < nb) {
                Ab[lu_ab(L, g, n + cb, rb)] = El<REAL>::one();
                ++cb;
            }
        }
    }
    const double tiny = 1e-13 * anorm;
    bad = bad || (anorm == 0.0);
    __syncthreads();                                   // the assembled matrix is in global memory, visible to the workgroup
    const bool interior = ty <= kl;
    const int rb = ty - kl - 1;                        // border row of a border thread
    int my_row = ty;                                   // interior: the row this thread holds
    E reg[FW_WMAX_T];                                    // reg[w]: the entry of my row in column j + w
#pragma unroll
    for (int w = 0; w < FW_WMAX_T; ++w) {
        reg[w] = El<REAL>::zero();
        if (w <= W) {
            if (interior) {
                if (my_row < n) reg[w] = Aw[lu_aw(L, g, my_row, kl - my_row + w)];     // column w of row r <= kl
            } else if (rb < nb && w < N) {
                reg[w] = Ab[lu_ab(L, g, w, rb)];
            }
        }
    }
    for (int j = 0; j < n; ++j) {
        const int imax = (j + kl < n) ? kl : (n - 1 - j);
        const int wmax = (j + W < N) ? W : (N - 1 - j);
        const int i = my_row - j;                      // my row's slot in the window (interior threads)
        const bool live = interior && my_row < n;      // then 0 <= i <= imax
        // ---- (A) pivot search: candidates in slot order
        if (interior) s_val[ty][tx] = live ? El<REAL>::abs2(reg[0]) : -1.0;
        __syncthreads();
        int p = 0;
        {
            double best = -1.0;
            int t = j % KL1;
            for (int q = 0; q <= imax; ++q) {
                const double m = s_val[t][tx];
                if (m > best) {
                    best = m;
                    p = q;
                }
                if (++t == KL1) t = 0;
            }
        }
        if (live && i == 0) L.piv[lu_pv(L, g, j)] = (unsigned char)p;
        // ---- (B) the pivot row (slot p) and the row it displaces (slot 0) go through LDS
        if (live && i == p) {
#pragma unroll
            for (int w = 0; w < FW_WMAX_T; ++w) {
                if (w <= W) s_piv[w * 64 + tx] = reg[w];
                if ((w & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (live && i == 0) {
#pragma unroll
            for (int w = 0; w < FW_WMAX_T; ++w) {
                if (w <= W) s_disp[w * 64 + tx] = reg[w];
                if ((w & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        E piv = s_piv[tx];
        if (!(El<REAL>::abs2(piv) > tiny * tiny)) {
            bad = true;
            piv = El<REAL>::one();
        }
        const E ip = El<REAL>::inv(piv);
        if (live && i == p && p != 0) {                // the old row j lives on in slot p
#pragma unroll
            for (int w = 0; w < FW_WMAX_T; ++w) {
                if (w <= W) reg[w] = s_disp[w * 64 + tx];
                if ((w & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- (C) multiplier and update of my row (slots 1 .. imax, border rows)
        const bool upd = (live && i >= 1) || (!interior && rb < nb);
        if (upd) {
            const E m = El<REAL>::mul(reg[0], ip);
            if (interior) Aw[lu_aw(L, g, my_row, kl - i)] = m;
            else Ab[lu_ab(L, g, j, rb)] = m;
            if (!El<REAL>::is_zero(m)) {
#pragma unroll
                for (int w = 1; w < FW_WMAX_T; ++w) {
                    if (w <= wmax) El<REAL>::fms(reg[w], m, s_piv[w * 64 + tx]);
                    if ((w & 7) == 7) __builtin_amdgcn_sched_barrier(0);       // at most 8 LDS reads in flight (registers)
                }
            }
        }
        // the finished U row j (reciprocal pivot in front), dealt out to the row threads
        for (int d = ty; d <= wmax; d += NT) Aw[lu_aw(L, g, j, kl + d)] = (d == 0) ? ip : s_piv[d * 64 + tx];
        // ---- advance the window
        if (interior && live && i == 0) {
            my_row = j + kl + 1;                       // the row entering the window takes this thread
            {
                const E *rowp = Aw + lu_aw(L, g, my_row < n ? my_row : 0, 0);         // one base, entry offsets added per load
#pragma unroll
                for (int w = 0; w < FW_WMAX_T; ++w) {
                    reg[w] = El<REAL>::zero();
                    if (w <= W && my_row < n) reg[w] = rowp[lu_eoff(L, w + L.kpad) - lu_eoff(L, L.kpad)];
                    if ((w & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int w = 0; w + 1 < FW_WMAX_T; ++w) reg[w] = reg[w + 1];
            reg[FW_WMAX_T - 1] = El<REAL>::zero();
            if (!interior && rb < nb) {
                // the column entering a border row's window
                const int cn = j + 1 + W;
#pragma unroll
                for (int w = 0; w < FW_WMAX_T; ++w)
                    if (w == W) reg[w] = (cn < N) ? Ab[lu_ab(L, g, cn, rb)] : El<REAL>::zero();
            }
        }
        // (no barrier: s_val / s_piv / s_disp are rewritten only after the next step's barriers)
    }
    // border rows: what is left of their window are the columns n .. N - 1
    if (!interior && rb < nb) {
#pragma unroll
        for (int w = 0; w < FW_WMAX_T; ++w)
            if (w < N - n) Ab[lu_ab(L, g, n + w, rb)] = reg[w];
    }
    __syncthreads();
    if (ty == 0 && nb > 0 && g < L.GL) {
        E *Sinv = (E *)L.scratch;   // [nb*nb][GL] workspace
        const long gs = g;
        for (int r = 0; r < nb; ++r)
            for (int cidx = 0; cidx < nb; ++cidx)
                Sinv[((long)r * nb + cidx) * G + gs] = (r == cidx) ? El<REAL>::one() : El<REAL>::zero();
        for (int k = 0; k < nb; ++k) {
            int p = k;
            double best = -1.0;
            for (int r = k; r < nb; ++r) {
                const double m = El<REAL>::abs2(Ab[lu_ab(L, g, n + k, r)]);
                if (m > best) {
                    best = m;
                    p = r;
                }
            }
            if (p != k) {
                for (int cidx = 0; cidx < nb; ++cidx) {
                    E *pa = Ab + lu_ab(L, g, n + cidx, k), *pb = Ab + lu_ab(L, g, n + cidx, p);
                    E t = *pa;
                    *pa = *pb;
                    *pb = t;
                    pa = Sinv + ((long)k * nb + cidx) * G + gs;
                    pb = Sinv + ((long)p * nb + cidx) * G + gs;
                    t = *pa;
                    *pa = *pb;
                    *pb = t;
                }
            }
            E piv = Ab[lu_ab(L, g, n + k, k)];
            if (!(El<REAL>::abs2(piv) > tiny * tiny)) {
                bad = true;
                piv = El<REAL>::one();
            }
            const E ip = El<REAL>::inv(piv);
            for (int cidx = 0; cidx < nb; ++cidx) {
                E *pa = Ab + lu_ab(L, g, n + cidx, k);
                *pa = El<REAL>::mul(*pa, ip);
                pa = Sinv + ((long)k * nb + cidx) * G + gs;
                *pa = El<REAL>::mul(*pa, ip);
            }
            for (int r = 0; r < nb; ++r) {
                if (r == k) continue;
                const E m = Ab[lu_ab(L, g, n + k, r)];
                if (El<REAL>::is_zero(m)) continue;
                for (int cidx = 0; cidx < nb; ++cidx) {
                    E *pt = Ab + lu_ab(L, g, n + cidx, r);
                    E t = *pt;
                    El<REAL>::fms(t, m, Ab[lu_ab(L, g, n + cidx, k)]);
                    *pt = t;
                    pt = Sinv + ((long)r * nb + cidx) * G + gs;
                    t = *pt;
                    El<REAL>::fms(t, m, Sinv[((long)k * nb + cidx) * G + gs]);
                    *pt = t;
                }
            }
        }
        for (int r = 0; r < nb; ++r)
            for (int cidx = 0; cidx < nb; ++cidx)
                Ab[lu_ab(L, g, n + cidx, r)] = Sinv[((long)r * nb + cidx) * G + gs];
    }
    // (every thread of a system saw the same pivots; the assembly flags were combined above)
    if (ty == 0 && g < L.GL) L.flag[g] = bad ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// solve: forward sweep (row interchanges, band multipliers, border multipliers), Schur block,
// backward sweep with a register window of the last W solution entries.
// ------------------------------------------------------------------------------------------------
// The solve is split into a forward and a backward kernel (different register needs), both software
// pipelined: everything row j+1 needs (pivot, multipliers, right-hand side / U row) is requested
// before row j is consumed, so each row costs one overlapped memory latency instead of several
// serialized ones.  Permutations and grading codes are staged in LDS.
template <int NF, bool REAL, int KLT, int NBT>
__global__ void __launch_bounds__(256)
solve_forward_kernel(PencilDev P, LuDev L, const RhsSrc rhs, double *__restrict__ xout) {
    typedef typename El<REAL>::T E;
    extern __shared__ int s_lds[];
    const int N = L.N;
    int *s_perm = s_lds;                                    // rowperm, then colperm of the border
    int *s_perm2 = s_lds + (L.pair ? N + L.nb : 0);         // the partner's (aliases s_perm when unpaired)
    unsigned char *s_code = (unsigned char *)(s_perm2 + N + L.nb);
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        s_perm[i] = L.rowperm[i];
        if (L.pair) s_perm2[i] = L.rowperm2[i];
        s_code[i] = REAL ? L.row_code[i] : 0;
    }
    for (int i = threadIdx.x; i < L.nb; i += blockDim.x) {
        s_perm[N + i] = L.colperm[L.n + i];
        if (L.pair) s_perm2[N + i] = L.colperm2[L.n + i];
        s_code[N + i] = REAL ? L.col_code[L.n + i] : 0;
    }
    __syncthreads();
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const SysId id = sys_id<REAL>(P, L, g);
    if (!id.ok) return;
    const int s = id.s;
    const CellCtx c = cell_ctx(P, id.cell);
    const long G = id.G;
    const long gl = id.gl;
    const int *my_perm = id.partner ? s_perm2 : s_perm;
    const bool conjq = id.partner && s == 1;
    const E *Aw = (const E *)L.Aw, *Ab = (const E *)L.Ab;
    const long plane = P.nx * P.ny;
    const int n = L.n, nb = L.nb, kl = L.kl;

    auto load_row = [&](int i) -> double2 {
        double2 v = load_sys<NF>(rhs, plane, my_perm[i], P, c, s);
        if (conjq) v.y = -v.y;
        if (REAL) {
            const unsigned char code = s_code[i];
            if ((code & 2) && s == 1) v = make_double2(-v.x, -v.y);
            if (code & 1) v = make_double2(v.y, -v.x);
        }
        return v;
    };

    double2 w[KLT + 1];
#pragma unroll
    for (int d = 0; d <= KLT; ++d) {
        w[d] = make_double2(0.0, 0.0);
        if (d <= kl && d < n) w[d] = load_row(d);
    }
    double2 gb[NBT];
#pragma unroll
    for (int rb = 0; rb < NBT; ++rb) {
        gb[rb] = make_double2(0.0, 0.0);
        if (rb < nb) gb[rb] = load_row(n + rb);
    }
    // prefetch registers for the next row
    int p_nx = 0;
    E m_nx[KLT], ab_nx[NBT];
    double2 r_nx = make_double2(0.0, 0.0);
#pragma unroll
    for (int i = 0; i < KLT; ++i) m_nx[i] = El<REAL>::zero();
#pragma unroll
    for (int rb = 0; rb < NBT; ++rb) ab_nx[rb] = El<REAL>::zero();
    auto prefetch = [&](int j) {
        p_nx = L.piv[lu_pv(L, gl, j)];
#pragma unroll
        for (int i = 1; i <= KLT; ++i)
            if (i <= kl && j + i < n) m_nx[i - 1] = Aw[lu_aw(L, gl, j + i, kl - i)];
#pragma unroll
        for (int rb = 0; rb < NBT; ++rb)
            if (rb < nb) ab_nx[rb] = Ab[lu_ab(L, gl, j, rb)];
        const int nxt = j + kl + 1;
        r_nx = (nxt < n) ? load_row(nxt) : make_double2(0.0, 0.0);
    };
    if (n > 0) prefetch(0);
    for (int j = 0; j < n; ++j) {
        const int p = p_nx;
        E m[KLT], ab[NBT];
#pragma unroll
        for (int i = 0; i < KLT; ++i) m[i] = m_nx[i];
#pragma unroll
        for (int rb = 0; rb < NBT; ++rb) ab[rb] = ab_nx[rb];
        const double2 rnew = r_nx;
        if (j + 1 < n) prefetch(j + 1);
        double2 yj = w[0];
#pragma unroll
        for (int d = 1; d <= KLT; ++d) {
            if (d == p) {
                yj = w[d];
                w[d] = w[0];
            }
        }
        L.scratch[(long)j * G + g] = yj;
#pragma unroll
        for (int i = 1; i <= KLT; ++i)
            if (i <= kl && j + i < n) El<REAL>::fms2(w[i], m[i - 1], yj);
#pragma unroll
        for (int rb = 0; rb < NBT; ++rb)
            if (rb < nb) El<REAL>::fms2(gb[rb], ab[rb], yj);
#pragma unroll
        for (int d = 0; d < KLT; ++d) w[d] = w[d + 1];
        w[KLT] = make_double2(0.0, 0.0);
#pragma unroll
        for (int d = 0; d <= KLT; ++d)
            if (d == kl) w[d] = rnew;
    }
    // ---- Schur block: z = Sinv * gb ; border unknown r is logical column n + r
#pragma unroll
    for (int r = 0; r < NBT; ++r) {
        if (r < nb) {
            double2 acc = make_double2(0.0, 0.0);
#pragma unroll
            for (int cidx = 0; cidx < NBT; ++cidx)
                if (cidx < nb) El<REAL>::fma2(acc, Ab[lu_ab(L, gl, n + cidx, r)], gb[cidx]);
            L.scratch[(long)(n + r) * G + g] = acc;     // graded value for the backward sweep
            double2 v = acc;
            if (REAL) {
                const unsigned char code = s_code[N + r];
                if (code & 1) v = make_double2(-v.y, v.x);
                if ((code & 2) && s == 1) v = make_double2(-v.x, -v.y);
            }
            if (conjq) v.y = -v.y;
            store_sys<NF>(xout, plane, my_perm[N + r], P, c, s, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Forward sweep of the real-graded two-axis path, one thread per system, written for the instruction stream: the general
// kernel above spends ~1500 instructions per row (per-entry guards, 64-bit index arithmetic per load, scalar registers
// spilled into vector lanes, a select chain for the row interchange) on 38 useful multiply-adds and was bound by
// instruction issue, not by memory.  Here:
//  * the factor rows are padded (kpad zero entries in front, KLT zero rows behind): the KLT multipliers of column j are
//    loaded without guards from awl[j * BW64 + coff[i]] -- one per-lane base pointer, 32-bit uniform offsets;
//  * the register window holds rows j .. j + KLT: the new row always enters at the top (rows beyond j + kl are not touched
//    by the elimination until they come within kl of the pivot row: their multipliers are zero);
//  * the interchange's select chain only runs when some system of the wavefront interchanges at this step;
//  * two register sets alternate for the prefetched row (no copies), all loads of a row are issued before its first use.
// Same arithmetic, same order of operations per system as solve_forward_kernel<2, true, ...>.
// ------------------------------------------------------------------------------------------------
template <int KLT, int NBT, int MINW = 1>
__global__ void __launch_bounds__(256, MINW)
solve_forward_lean_kernel(PencilDev P, LuDev L, const RhsSrc rhs, double *__restrict__ xout) {
    constexpr int NF = 2;
    extern __shared__ int s_lds[];
    const int N = L.N, n = L.n, nb = L.nb;
    int *s_perm = s_lds;                                    // rowperm, then colperm of the border
    int *s_perm2 = s_lds + (L.pair ? N + nb : 0);           // the partner's (aliases s_perm when unpaired)
    unsigned char *s_code = (unsigned char *)(s_perm2 + N + nb);
    unsigned char *s_zero = s_code + N + nb;                // rows whose right-hand side is zero in every term: not read
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        s_perm[i] = L.rowperm[i];
        if (L.pair) s_perm2[i] = L.rowperm2[i];
        s_code[i] = L.row_code[i];
        s_zero[i] = rhs.zrow ? (rhs.zrow[L.rowperm[i]] && (!L.pair || rhs.zrow[L.rowperm2[i]])) : 0;
    }
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
        s_perm[N + i] = L.colperm[n + i];
        if (L.pair) s_perm2[N + i] = L.colperm2[n + i];
        s_code[N + i] = L.col_code[n + i];
    }
    __syncthreads();
    // independent diagonal blocks (LuDev::nsplit): this thread sweeps rows row0 .. row1 - 1 of system g
    // (Gp is a multiple of the workgroup size: the block index is uniform over the workgroup -- computed from blockIdx
    // alone so that the row counters and everything indexed by them stay in scalar registers)
    const int blk = L.nsplit > 1 ? __builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * blockDim.x) / L.Gp)) : 0;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x - (long)blk * L.Gp;
    const int row0 = blk * L.nh, row1 = row0 + L.nh;
    const SysId id = sys_id<true>(P, L, g);
    if (!id.ok) return;
    const int s = id.s;
    const CellCtx c = cell_ctx(P, id.cell);
    const long G = id.G;
    const int *my_perm = id.partner ? s_perm2 : s_perm;
    const bool conjq = id.partner && s == 1;
    const long plane = P.nx * P.ny;
    // running per-lane pointers, advanced by uniform strides once per row:
    //   mp[i]: multiplier L(j + i + 1, j) = row j + i + 1, entry KLT - i - 1;  abp: border multipliers of column j (nb entries,
    //   512 B apart: immediate offsets);  pvp: the interchange of step j;  scr: y_j of this system
    // (one per-lane base + a running 32-bit byte offset + KLT wave-uniform deltas: 3 vector registers instead of 2 KLT)
    const char *const awl = (const char *)((const double *)L.Aw + lu_aw(L, id.gl, row0, -L.kpad));   // this lane's block, first row, entry 0
    unsigned mo = 0;
    unsigned md[KLT];
    {
        const long BW64 = (long)L.BW << 6;
#pragma unroll
        for (int i = 0; i < KLT; ++i) md[i] = (unsigned)(((long)(i + 1) * BW64 + lu_eoff(L, KLT - i - 1)) * 8);
    }
    const double *abp = (const double *)L.Ab + lu_ab(L, id.gl, row0, 0);
    const unsigned char *pvp = L.piv + lu_pv(L, id.gl, row0);
    double2 *scr = L.scratch + (long)row0 * G + g;
    const unsigned row_step8 = (unsigned)L.BW << 9;              // bytes between band rows of a block (32-bit: a block is < 4 GB)
    const long ab_step = (long)nb << 6;

    auto load_row = [&](int i) -> double2 {
        if (s_zero[i]) return make_double2(0.0, 0.0);           // wave-uniform
#ifdef DDH_SWEEP_ABLATE
        if (rhs.abl & 1) {                                      // right-hand-side terms from lane-contiguous addresses
            double2 acc = make_double2(0.0, 0.0);
#pragma unroll
            for (int t = 0; t < RHS_MAX; ++t)
                if (t < rhs.n) {
                    const double2 q = *reinterpret_cast<const double2 *>(rhs.p[t] + (long)i * plane + 2 * g);
                    acc.x += rhs.a[t] * q.x;
                    acc.y += rhs.a[t] * q.y;
                }
            return acc;
        }
#endif
        double2 v = load_sys<NF>(rhs, plane, my_perm[i], P, c, s);
        if (conjq) v.y = -v.y;
        const unsigned char code = s_code[i];
        if ((code & 2) && s == 1) v = make_double2(-v.x, -v.y);
        if (code & 1) v = make_double2(v.y, -v.x);
        return v;
    };

    // the window (rows 0 .. KLT) and the border rows are shifted in by rolled loops: one inlined copy of the row load each
    double2 w[KLT + 1];
#pragma unroll
    for (int d = 0; d <= KLT; ++d) w[d] = make_double2(0.0, 0.0);
#pragma unroll 1
    for (int t = 0; t <= KLT; ++t) {
        const double2 v = (row0 + t < row1) ? load_row(row0 + t) : make_double2(0.0, 0.0);
#pragma unroll
        for (int d = 0; d < KLT; ++d) w[d] = w[d + 1];
        w[KLT] = v;
    }
    double2 gb[NBT];
#pragma unroll
    for (int rb = 0; rb < NBT; ++rb) gb[rb] = make_double2(0.0, 0.0);
#pragma unroll 1
    for (int t = 0; t < NBT; ++t) {
        const double2 v = (t < nb && blk == 0) ? load_row(n + t) : make_double2(0.0, 0.0);   // (block 0 carries the border's right-hand side)
#pragma unroll
        for (int rb = 0; rb + 1 < NBT; ++rb) gb[rb] = gb[rb + 1];
        gb[NBT - 1] = v;
    }
    double mA[KLT], mB[KLT], abA[NBT], abB[NBT];
    int pA = 0, pB = 0;
    double2 rA = make_double2(0.0, 0.0), rB = rA;
#pragma unroll
    for (int i = 0; i < KLT; ++i) mA[i] = mB[i] = 0.0;
#pragma unroll
    for (int rb = 0; rb < NBT; ++rb) abA[rb] = abB[rb] = 0.0;
    const bool full_border = (nb == NBT);
    // loads of the NEXT row (the pointers stand at it); called for rows 0, 1, 2, ... in order
    int jn = row0;
    auto prefetch = [&](double *m, double *ab, int &p, double2 &r) {
        p = *pvp;
#pragma unroll
        for (int i = 0; i < KLT; ++i) m[i] = *reinterpret_cast<const double *>(awl + (mo + md[i]));
#ifdef DDH_SWEEP_ABLATE
        if (rhs.abl & 4) {                                      // one multiplier load per row
#pragma unroll
            for (int i = 1; i < KLT; ++i) m[i] = m[0] * 0.5;
        }
#endif
        if (full_border) {
#pragma unroll
            for (int rb = 0; rb < NBT; ++rb) ab[rb] = abp[rb << 6];
        } else {
#pragma unroll
            for (int rb = 0; rb < NBT; ++rb)
                if (rb < nb) ab[rb] = abp[rb << 6];
        }
        __builtin_amdgcn_sched_barrier(0);      // every factor load of the row is in flight before anything waits
        // (the right-hand-side row follows: its value is not needed before the end of the next step)
        const int nxt = jn + KLT + 1;
        r = make_double2(0.0, 0.0);
        if (nxt < row1) r = load_row(nxt);
        mo += row_step8;
        abp += ab_step;
        pvp += 64;
        ++jn;
    };
    auto step = [&](const double *m, const double *ab, int p, double2 rnew) {
        double2 yj = w[0];
#pragma unroll
        for (int d = 1; d <= KLT; ++d) {
            if (d == p) {
                yj = w[d];
                w[d] = w[0];
            }
        }
        *scr = yj;
        scr += G;
#pragma unroll
        for (int i = 1; i <= KLT; ++i) {
            w[i].x -= m[i - 1] * yj.x;
            w[i].y -= m[i - 1] * yj.y;
        }
#pragma unroll
        for (int rb = 0; rb < NBT; ++rb) {
            gb[rb].x -= ab[rb] * yj.x;
            gb[rb].y -= ab[rb] * yj.y;
        }
#pragma unroll
        for (int d = 0; d < KLT; ++d) w[d] = w[d + 1];
        w[KLT] = rnew;
    };
    // (a second row of loads in flight per system was measured: 4.8 ms instead of 4.4 ms at 512^2 pencils)
    if (row1 > row0) prefetch(mB, abB, pB, rB);
#pragma unroll 1
    for (int j = row0; j < row1; ++j) {
#pragma unroll
        for (int i = 0; i < KLT; ++i) mA[i] = mB[i];
#pragma unroll
        for (int rb = 0; rb < NBT; ++rb) abA[rb] = abB[rb];
        pA = pB;
        rA = rB;
        if (j + 1 < row1) prefetch(mB, abB, pB, rB);
        step(mA, abA, pA, rA);
    }
    if (L.nsplit > 1) {
        // the border rows collect contributions of every block: partial sums, finished by border_finish_kernel
#pragma unroll
        for (int r = 0; r < NBT; ++r)
            if (r < nb) L.scratch[(long)(n + nb + blk * nb + r) * G + g] = gb[r];
        return;
    }
    // ---- Schur block: z = Sinv * gb ; border unknown r is logical column n + r
    const double *Ab = (const double *)L.Ab;
#pragma unroll
    for (int r = 0; r < NBT; ++r) {
        if (r < nb) {
            double2 acc = make_double2(0.0, 0.0);
#pragma unroll
            for (int cidx = 0; cidx < NBT; ++cidx)
                if (cidx < nb) El<true>::fma2(acc, Ab[lu_ab(L, id.gl, n + cidx, r)], gb[cidx]);
            L.scratch[(long)(n + r) * G + g] = acc;     // graded value for the backward sweep
            double2 v = acc;
            const unsigned char code = s_code[N + r];
            if (code & 1) v = make_double2(-v.y, v.x);
            if ((code & 2) && s == 1) v = make_double2(-v.x, -v.y);
            if (conjq) v.y = -v.y;
            store_sys<NF>(xout, plane, my_perm[N + r], P, c, s, v);
        }
    }
}

template <int KLT, int NBT, int PD, int NT>       // NT: right-hand-side terms at most (rhs.n <= NT)
__global__ void __launch_bounds__(256, 1)
solve_forward_deep_kernel(PencilDev P, LuDev L, const RhsSrc rhs, double *__restrict__ xout) {
    constexpr int NF = 2;
    extern __shared__ int s_lds[];
    const int N = L.N, n = L.n, nb = L.nb;
    int *s_perm = s_lds;                                    // rowperm, then colperm of the border
    int *s_perm2 = s_lds + (L.pair ? N + nb : 0);           // the partner's (aliases s_perm when unpaired)
    unsigned char *s_code = (unsigned char *)(s_perm2 + N + nb);
    unsigned char *s_zero = s_code + N + nb;                // rows whose right-hand side is zero in every term: not read
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        s_perm[i] = L.rowperm[i];
        if (L.pair) s_perm2[i] = L.rowperm2[i];
        s_code[i] = L.row_code[i];
        s_zero[i] = rhs.zrow ? (rhs.zrow[L.rowperm[i]] && (!L.pair || rhs.zrow[L.rowperm2[i]])) : 0;
    }
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
        s_perm[N + i] = L.colperm[n + i];
        if (L.pair) s_perm2[N + i] = L.colperm2[n + i];
        s_code[N + i] = L.col_code[n + i];
    }
    __syncthreads();
    // independent diagonal blocks (LuDev::nsplit): this thread sweeps rows row0 .. row1 - 1 of system g
    // (Gp is a multiple of the workgroup size: the block index is uniform over the workgroup -- computed from blockIdx
    // alone so that the row counters and everything indexed by them stay in scalar registers)
    const int blk = L.nsplit > 1 ? __builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * blockDim.x) / L.Gp)) : 0;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x - (long)blk * L.Gp;
    const int row0 = blk * L.nh, row1 = row0 + L.nh;
    const SysId id = sys_id<true>(P, L, g);
    if (!id.ok) return;
    const int s = id.s;
    const CellCtx c = cell_ctx(P, id.cell);
    const long G = id.G;
    const int *my_perm = id.partner ? s_perm2 : s_perm;
    const bool conjq = id.partner && s == 1;
    const long plane = P.nx * P.ny;
    // running per-lane pointers, advanced by uniform strides once per row:
    //   mp[i]: multiplier L(j + i + 1, j) = row j + i + 1, entry KLT - i - 1;  abp: border multipliers of column j (nb entries,
    //   512 B apart: immediate offsets);  pvp: the interchange of step j;  scr: y_j of this system
    // (one per-lane base + a running 32-bit byte offset + KLT wave-uniform deltas: 3 vector registers instead of 2 KLT)
    const char *const awl = (const char *)((const double *)L.Aw + lu_aw(L, id.gl, row0, -L.kpad));   // this lane's block, first row, entry 0
    unsigned mo = 0;
    unsigned md[KLT];
    {
        const long BW64 = (long)L.BW << 6;
#pragma unroll
        for (int i = 0; i < KLT; ++i) md[i] = (unsigned)(((long)(i + 1) * BW64 + lu_eoff(L, KLT - i - 1)) * 8);
    }
    const double *abp = (const double *)L.Ab + lu_ab(L, id.gl, row0, 0);
    const unsigned char *pvp = L.piv + lu_pv(L, id.gl, row0);
    double2 *scr = L.scratch + (long)row0 * G + g;
    const unsigned row_step8 = (unsigned)L.BW << 9;              // bytes between band rows of a block (32-bit: a block is < 4 GB)
    const long ab_step = (long)nb << 6;

    auto load_row = [&](int i) -> double2 {
        if (s_zero[i]) return make_double2(0.0, 0.0);           // wave-uniform
#ifdef DDH_SWEEP_ABLATE
        if (rhs.abl & 1) {                                      // right-hand-side terms from lane-contiguous addresses
            double2 acc = make_double2(0.0, 0.0);
#pragma unroll
            for (int t = 0; t < RHS_MAX; ++t)
                if (t < rhs.n) {
                    const double2 q = *reinterpret_cast<const double2 *>(rhs.p[t] + (long)i * plane + 2 * g);
                    acc.x += rhs.a[t] * q.x;
                    acc.y += rhs.a[t] * q.y;
                }
            return acc;
        }
#endif
        double2 v = load_sys<NF>(rhs, plane, my_perm[i], P, c, s);
        if (conjq) v.y = -v.y;
        const unsigned char code = s_code[i];
        if ((code & 2) && s == 1) v = make_double2(-v.x, -v.y);
        if (code & 1) v = make_double2(v.y, -v.x);
        return v;
    };

    // the window (rows 0 .. KLT) and the border rows are shifted in by rolled loops: one inlined copy of the row load each
    double2 w[KLT + 1];
#pragma unroll
    for (int d = 0; d <= KLT; ++d) w[d] = make_double2(0.0, 0.0);
#pragma unroll 1
    for (int t = 0; t <= KLT; ++t) {
        const double2 v = (row0 + t < row1) ? load_row(row0 + t) : make_double2(0.0, 0.0);
#pragma unroll
        for (int d = 0; d < KLT; ++d) w[d] = w[d + 1];
        w[KLT] = v;
    }
    double2 gb[NBT];
#pragma unroll
    for (int rb = 0; rb < NBT; ++rb) gb[rb] = make_double2(0.0, 0.0);
#pragma unroll 1
    for (int t = 0; t < NBT; ++t) {
        const double2 v = (t < nb && blk == 0) ? load_row(n + t) : make_double2(0.0, 0.0);   // (block 0 carries the border's right-hand side)
#pragma unroll
        for (int rb = 0; rb + 1 < NBT; ++rb) gb[rb] = gb[rb + 1];
        gb[NBT - 1] = v;
    }
    // PD register sets: the factor data of column j and the raw right-hand-side terms of the row that enters the window
    // at step j, requested PD steps ahead (set = (j - row0) % PD: the loop below is unrolled PD-fold)
    double m[PD][KLT], ab[PD][NBT];
    int pv[PD];
    double2 raw[PD][NT];
#pragma unroll
    for (int d = 0; d < PD; ++d) {
        pv[d] = 0;
#pragma unroll
        for (int i = 0; i < KLT; ++i) m[d][i] = 0.0;
#pragma unroll
        for (int rb = 0; rb < NBT; ++rb) ab[d][rb] = 0.0;
#pragma unroll
        for (int t = 0; t < NT; ++t) raw[d][t] = make_double2(0.0, 0.0);
    }
    // this thread's offset inside a row of the term vectors (tile-major or natural layout)
    const long cell_off = rhs.tiled ? tile_offset(2 * c.mx + s, 2 * c.my, P.ny) : (2 * c.mx + s) * P.ny + 2 * c.my;
    // BRANCH-FREE: a uniform branch around any of these loads makes the number of loads in flight depend on the path, and
    // the compiler then waits for the smaller count -- i.e. drains the pipeline at every step.  Hence: the border is full
    // (nb == NBT, checked at the launch), terms beyond rhs.n re-read term 0 (coefficient 0), rows that are structurally
    // zero or past the end of the block are read anyway (and discarded in `entering`).
    const double *pt[NT];
    double at[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        pt[t] = t < rhs.n ? rhs.p[t] : rhs.p[0];
        at[t] = t < rhs.n ? rhs.a[t] : 0.0;
    }
    auto request = [&](int jrow, double *mm, double *aa, int &p, double2 *q) {
        // (past the end of the block: the last row again -- never used, but the addresses stay inside the block)
        const unsigned jr = (unsigned)((jrow < row1 ? jrow : row1 - 1) - row0);
        const unsigned mo_j = jr * row_step8;
        p = pvp[(long)jr << 6];
#pragma unroll
        for (int i = 0; i < KLT; ++i) mm[i] = *reinterpret_cast<const double *>(awl + (mo_j + md[i]));
        const double *abj = abp + (long)jr * ab_step;
#pragma unroll
        for (int rb = 0; rb < NBT; ++rb) aa[rb] = abj[rb << 6];
        const int nxt = jrow + KLT + 1;
        const long off = (long)my_perm[nxt < row1 ? nxt : row1 - 1] * plane + cell_off;
#pragma unroll
        for (int t = 0; t < NT; ++t) q[t] = *reinterpret_cast<const double2 *>(pt[t] + off);
        __builtin_amdgcn_sched_barrier(0);      // the requests stay here (see solve_backward_deep_kernel)
    };
    // the row that enters the window at step jrow, from its raw terms: the arithmetic of load_sys / load_row, in their order
    auto entering = [&](int jrow, const double2 *q) -> double2 {
        const int nxt = jrow + KLT + 1;
        if (nxt >= row1 || s_zero[nxt]) return make_double2(0.0, 0.0);
        double2 v;
        if (rhs.n == 1 && !rhs.tiled) {
            const double2 mine = q[0];
            double2 other;
            other.x = __shfl_xor(mine.x, 1);
            other.y = __shfl_xor(mine.y, 1);
            v = s == 0 ? make_double2(mine.x - other.y, mine.y + other.x) : make_double2(other.x + mine.y, other.y - mine.x);
            if (rhs.a[0] != 1.0) v = make_double2(rhs.a[0] * v.x, rhs.a[0] * v.y);
        } else {
            double2 mine = make_double2(0.0, 0.0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                mine.x += at[t] * q[t].x;
                mine.y += at[t] * q[t].y;
            }
            double2 other;
            other.x = __shfl_xor(mine.x, 1);
            other.y = __shfl_xor(mine.y, 1);
            v = s == 0 ? make_double2(mine.x - other.y, mine.y + other.x) : make_double2(other.x + mine.y, other.y - mine.x);
        }
        if (conjq) v.y = -v.y;
        const unsigned char code = s_code[nxt];
        if ((code & 2) && s == 1) v = make_double2(-v.x, -v.y);
        if (code & 1) v = make_double2(v.y, -v.x);
        return v;
    };
    auto step = [&](const double *mm, const double *aa, int p, double2 rnew) {
        double2 yj = w[0];
#pragma unroll
        for (int d = 1; d <= KLT; ++d) {
            if (d == p) {
                yj = w[d];
                w[d] = w[0];
            }
        }
        *scr = yj;
        scr += G;
#pragma unroll
        for (int i = 1; i <= KLT; ++i) {
            w[i].x -= mm[i - 1] * yj.x;
            w[i].y -= mm[i - 1] * yj.y;
        }
#pragma unroll
        for (int rb = 0; rb < NBT; ++rb) {
            gb[rb].x -= aa[rb] * yj.x;
            gb[rb].y -= aa[rb] * yj.y;
        }
#pragma unroll
        for (int d = 0; d < KLT; ++d) w[d] = w[d + 1];
        w[KLT] = rnew;
    };
    // (requests are unconditional: the factor rows are padded with KLT zero rows and rows_aw >= n + KLT, the right-hand-side
    //  row is guarded inside; a branch around them would make the compiler wait for everything, see the backward kernel)
#pragma unroll
    for (int d = 0; d < PD; ++d) request(row0 + d, m[d], ab[d], pv[d], raw[d]);
    int j = row0;
    for (; j + PD - 1 < row1; j += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const double2 rnew = entering(j + d, raw[d]);
            step(m[d], ab[d], pv[d], rnew);
            request(j + d + PD, m[d], ab[d], pv[d], raw[d]);
        }
    }
#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (j + d < row1) step(m[d], ab[d], pv[d], entering(j + d, raw[d]));
    if (L.nsplit > 1) {
        // the border rows collect contributions of every block: partial sums, finished by border_finish_kernel
#pragma unroll
        for (int r = 0; r < NBT; ++r)
            if (r < nb) L.scratch[(long)(n + nb + blk * nb + r) * G + g] = gb[r];
        return;
    }
    // ---- Schur block: z = Sinv * gb ; border unknown r is logical column n + r
    const double *Ab = (const double *)L.Ab;
#pragma unroll
    for (int r = 0; r < NBT; ++r) {
        if (r < nb) {
            double2 acc = make_double2(0.0, 0.0);
#pragma unroll
            for (int cidx = 0; cidx < NBT; ++cidx)
                if (cidx < nb) El<true>::fma2(acc, Ab[lu_ab(L, id.gl, n + cidx, r)], gb[cidx]);
            L.scratch[(long)(n + r) * G + g] = acc;     // graded value for the backward sweep
            double2 v = acc;
            const unsigned char code = s_code[N + r];
            if (code & 1) v = make_double2(-v.y, v.x);
            if ((code & 2) && s == 1) v = make_double2(-v.x, -v.y);
            if (conjq) v.y = -v.y;
            store_sys<NF>(xout, plane, my_perm[N + r], P, c, s, v);
        }
    }
}

#ifdef DDH_SWEEP_ABLATE
__constant__ int c_abl;          // timing ablations of the backward sweep (DDH_ABL): 2 lane-contiguous stores, 8 one factor load per row
#endif
template <int NF, int WT, bool REAL, bool PREF, bool PFUSE = false, int DBG = 0, int MINW = 1>
__global__ void __launch_bounds__(256, MINW)
solve_backward_kernel(PencilDev P, LuDev L, double *__restrict__ xout, const double *__restrict__ pband,
                      const unsigned char *__restrict__ skip) {
    typedef typename El<REAL>::T E;
    extern __shared__ int s_lds[];
    const int n = L.n, nb = L.nb, kl = L.kl, W = L.W;
    int *s_perm = s_lds;
    int *s_perm2 = s_lds + (L.pair ? n : 0);
    unsigned char *s_code = (unsigned char *)(s_perm2 + n);
    unsigned char *s_skip = s_code + n;                 // unknowns the caller does not need: not stored
    unsigned char *s_nq = s_skip + n;                   // entry pairs of U row j that can hold data (LuDev::wrow)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        s_perm[i] = L.colperm[i];
        if (L.pair) s_perm2[i] = L.colperm2[i];
        s_code[i] = REAL ? L.col_code[i] : 0;
        s_skip[i] = skip ? (skip[L.colperm[i]] && (!L.pair || skip[L.colperm2[i]])) : 0;
        if (REAL) {
            const int full = (WT + 2) / 2;
            const int need = L.wrow ? (L.wrow[i] + 2) >> 1 : full;
            s_nq[i] = (unsigned char)(need < full ? need : full);
        }
    }
    __syncthreads();
    // independent diagonal blocks (LuDev::nsplit; only launched that way for REAL): rows row1 - 1 .. row0 of system g
    // (uniform over the workgroup: from blockIdx alone, so that the row counter stays in scalar registers)
    const int blk = (REAL && L.nsplit > 1) ? __builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * blockDim.x) / L.Gp)) : 0;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x - (long)blk * L.Gp;
    const int row0 = (REAL && L.nsplit > 1) ? blk * L.nh : 0, row1 = (REAL && L.nsplit > 1) ? row0 + L.nh : n;
    const SysId id = sys_id<REAL>(P, L, g);
    if constexpr ((DBG & 64) != 0) {                 // only the waves solve_backward_ring_kernel leaves out (ring_wave)
        if (ring_wave(id, (int)(threadIdx.x & 63))) return;
    }
    if (!id.ok) return;
    const int s = id.s;
    const CellCtx c = cell_ctx(P, id.cell);
    const long G = id.G;
    const long gl = id.gl;
    const int *my_perm = id.partner ? s_perm2 : s_perm;
    const bool conjq = id.partner && s == 1;
    const E *Aw = (const E *)L.Aw;
    const long plane = P.nx * P.ny;

    double2 win[WT];   // win[d] = x[j+1+d] (graded)
#pragma unroll
    for (int d = 0; d < WT; ++d) {
        win[d] = make_double2(0.0, 0.0);
        if (d < nb && row1 == n) win[d] = L.scratch[(long)(n + d) * G + g];   // border columns follow the LAST band rows
    }
    E ua[WT + 1], ub[WT + 1];
    double pa[PBW], pb[PBW];        // the row of the recombination band (PFUSE): uniform, loaded with the factor row
    double2 ya = make_double2(0.0, 0.0), yb = ya;
#pragma unroll
    for (int d = 0; d < PBW; ++d) pa[d] = pb[d] = 0.0;
#pragma unroll
    for (int d = 0; d <= WT; ++d) ua[d] = ub[d] = El<REAL>::zero();
    // FULL: every row stores WT + 1 entries from the diagonal on (zero padded beyond the band, factor_impl) -- no guards
    // in the row loops: every guard was a scalar branch or a select per entry, several times the useful work of a row.
    constexpr bool FULL = REAL;
    const E *const ur0 = Aw + lu_aw(L, gl, 0, kl);     // row 0, diagonal; a row further is BW * 64 elements further
    const long ur_stride = (long)L.BW << 6;
    const double2 *const y0 = L.scratch + g;
    // Real factors: the wave's addresses are  (uniform base of the row) + (32-bit byte offset of the lane): one vector
    // register instead of a 64-bit pointer per array and per out-of-range immediate offset.  The factorizations of a
    // wave's lanes are consecutive (sys_id), i.e. at most two adjacent 64-blocks of the storage: the first lane's
    // address is the smallest and the others lie within 2 * rows_aw * BW * 512 bytes of it (checked at factor time).
    // real factors: wave-uniform buffer addressing (wave_rsrc) of the U rows and of the scratch vector; the uniform
    // offsets are row * (bytes per row): < 4 GiB for both (rows_aw * BW * 512 per block, (n + nb) * G * 16)
    const __amdgpu_buffer_rsrc_t ur_rs = wave_rsrc(ur0);
    const unsigned ur_lane = wave_lane_off(ur0);
    const __amdgpu_buffer_rsrc_t y_rs = wave_rsrc(y0);
    const unsigned y_lane = wave_lane_off(y0);
    const unsigned ur_row8 = (unsigned)L.BW << 9, y_row16 = (unsigned)(G * (long)sizeof(double2));
    auto fetch = [&](int j, E *u, double2 &y, double *pr) {
        if (DBG & 8) y = make_double2(1.0, 2.0);
        else if constexpr (REAL) y = bload16(y_rs, y_lane, (unsigned)j * y_row16);
        else y = y0[(long)j * G];
        if (PFUSE) {
            const double *prow = pband + (long)j * PBW;
#pragma unroll
            for (int d = 0; d < PBW; ++d) pr[d] = prow[d];
        }
        const E *Ur = ur0 + (long)j * ur_stride;
        if constexpr (REAL && (DBG & 128) != 0) {
            // TIMING ONLY (DDH_BWD_DBG=128, results are not a solve): the factor rows addressed as if the storage were
            // [row][block of 64][entry][lane] instead of [block][row][entry][lane] -- at any time the whole chip then reads
            // one contiguous 6 MB region per row instead of 9 KB pieces of a thousand streams 15 MB apart
            const E *rm = (const E *)L.Aw + ((((long)j * L.nblk + (gl >> 6)) * L.BW) << 6) + lu_eoff(L, kl + L.kpad) + ((gl & 63) << 1);
            const __amdgpu_buffer_rsrc_t rs_j = wave_rsrc(rm);
            const unsigned ln_j = wave_lane_off(rm);
#pragma unroll
            for (int q = 0; 2 * q <= WT; ++q) {
                const double2 uu = bload16(rs_j, ln_j, (unsigned)(q << 10));
                u[2 * q] = uu.x;
                if (2 * q + 1 <= WT) u[2 * q + 1] = uu.y;
            }
        } else if constexpr (REAL) {
            // pair-packed rows (LuDev::pk; the diagonal sits at an even entry): 16-byte loads, two entries each
            const unsigned urow = (unsigned)j * ur_row8;                                 // uniform
            // pairs beyond the measured fill of this row are exact zeros in every factorization: not loaded (wave-uniform;
            // the first NQMIN pairs are loaded unconditionally -- nearly every row needs them, and every branch costs)
            constexpr int NQMIN = (WT >= 13) ? 6 : (WT + 2) / 2;
            const int nq = __builtin_amdgcn_readfirstlane((int)s_nq[j]);
#pragma unroll
            for (int q = 0; 2 * q <= WT; ++q) {
#ifdef DDH_SWEEP_ABLATE
                const bool one = (c_abl & 8) && q > 0;
#else
                constexpr bool one = false;
#endif
                double2 uu = make_double2(0.0, 0.0);
                if (((DBG & 1) && q > 0) || one) uu = make_double2(u[0] * 0.5, u[0] * 0.25);
                else if (q < NQMIN || q < nq) uu = bload16(ur_rs, ur_lane, urow + (unsigned)(q << 10));
                u[2 * q] = uu.x;
                if (2 * q + 1 <= WT) u[2 * q + 1] = uu.y;
            }
        } else {
#pragma unroll
            for (int d = 0; d <= WT; ++d)
                if (FULL || d <= W) u[d] = Ur[(long)d << 6];
        }
        // all loads of the row are in flight before the first use: left alone the scheduler trades them for registers
        // (4 outstanding loads, one memory round trip per group -- 9 per row instead of 1)
        __builtin_amdgcn_sched_barrier(0);
    };
    auto emit = [&](int j, double2 v) {
        if (s_skip[j]) return;                           // wave-uniform
        if (REAL) {
            const unsigned char code = s_code[j];
            if (code & 1) v = make_double2(-v.y, v.x);
            if ((code & 2) && s == 1) v = make_double2(-v.x, -v.y);
        }
        if (conjq) v.y = -v.y;
        if ((DBG & 4) && v.x != 1.2345e300) return;
        if (DBG & 32) { *reinterpret_cast<double2 *>(xout + (long)j * plane + 2 * g) = v; return; }
#ifdef DDH_SWEEP_ABLATE
        if (c_abl & 2) { *reinterpret_cast<double2 *>(xout + (long)j * plane + 2 * g) = v; return; }
#endif
        store_sys<NF>(xout, plane, my_perm[j], P, c, s, v);
    };
    // rows are processed in pairs; the register window is shifted once per pair (by two).  With PFUSE the emitted value
    // is the recombined unknown x_j = y_j + sum_d P[j, j + d] y_(j + d) (the window already holds y_(j+1..)): the
    // band coefficients are the same for every system -> scalar loads.
    auto row_even = [&](int j, const E *u, double2 y, const double *pr) -> double2 {
        double2 acc = y;
        if constexpr ((DBG & 2) != 0) {
            int o = 0;
#pragma unroll
            for (int d = 0; d < WT; ++d) o |= __double2loint(u[d + 1]);
            acc.x += (double)o * win[0].x;
        } else {
#pragma unroll
        for (int d = 0; d < WT; ++d)
            if (FULL || d < W) El<REAL>::fms2(acc, u[d + 1], win[d]);
        }
        const double2 xj = El<REAL>::mul2(acc, u[0]);   // reciprocal pivot stored on the diagonal
        double2 v = xj;
        if (PFUSE) {
#pragma unroll
            for (int d = 0; d < PBW; ++d)
                if (d < WT) { v.x += pr[d] * win[d].x; v.y += pr[d] * win[d].y; }
        }
        emit(j, v);
        return xj;
    };
    auto row_odd = [&](int j, const E *u, double2 y, double2 xprev, const double *pr) {
        double2 acc = y;
        El<REAL>::fms2(acc, u[1], xprev);
        if constexpr ((DBG & 2) != 0) {
            int o = 0;
#pragma unroll
            for (int d = 1; d < WT; ++d) o |= __double2loint(u[d + 1]);
            acc.x += (double)o * win[0].x;
        } else {
#pragma unroll
        for (int d = 1; d < WT; ++d)
            if (FULL || d < W) El<REAL>::fms2(acc, u[d + 1], win[d - 1]);
        }
        const double2 xj = El<REAL>::mul2(acc, u[0]);
        double2 v = xj;
        if (PFUSE) {
            v.x += pr[0] * xprev.x;
            v.y += pr[0] * xprev.y;
#pragma unroll
            for (int d = 1; d < PBW; ++d)
                if (d - 1 < WT) { v.x += pr[d] * win[d - 1].x; v.y += pr[d] * win[d - 1].y; }
        }
        emit(j, v);
        if constexpr ((DBG & 16) == 0) {
#pragma unroll
        for (int d = WT - 1; d > 1; --d) win[d] = win[d - 2];
        win[1] = xprev;
        }
        win[0] = xj;
    };
    int j = row1 - 1;
    if (PREF) {
        if (j >= row0) fetch(j, ua, ya, pa);
        while (j >= row0 + 1) {
            fetch(j - 1, ub, yb, pb);
            const double2 xe = row_even(j, ua, ya, pa);
            if (j - 2 >= row0) fetch(j - 2, ua, ya, pa);
            row_odd(j - 1, ub, yb, xe, pb);
            j -= 2;
        }
        if (j == row0) row_even(row0, ua, ya, pa);
    } else {
        // no register double-buffering: fewer VGPRs -> two waves per SIMD hide each other's latency
        while (j >= row0 + 1) {
            fetch(j, ua, ya, pa);
            const double2 xe = row_even(j, ua, ya, pa);
            fetch(j - 1, ua, ya, pa);
            row_odd(j - 1, ua, ya, xe, pa);
            j -= 2;
        }
        if (j == row0) {
            fetch(row0, ua, ya, pa);
            row_even(row0, ua, ya, pa);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward sweep for FEW systems (real-graded two-axis path): the loads of the next PD rows in flight, in registers.
//
// With fewer threads than two wavefronts per SIMD -- one rank's share of a sharded run (16 384 pencils at P = 4: one
// wave per SIMD), a 128^3 problem -- solve_backward_kernel is a chain of n dependent "load a row, wait one HBM round trip,
// 40 multiply-adds" steps with nothing to hide the round trips behind: ~1.5 us per row whatever the bandwidth
// (profiles/r6_rank_emulation.txt: 1.97 ms per solve at P = 8 against 0.73 ms for an eighth of the one-GPU solve).  One
// wave per SIMD owns the whole register file (512 VGPRs), so the rows of the next PD steps are simply requested PD steps
// ahead into PD register sets (the loop is unrolled PD-fold: every set is a compile-time index).  Same arithmetic in
// the same order: bit-identical to solve_backward_kernel.  Replaces the back substitution of the reference's per-pencil
// SuperLU solve (libraries/matsolvers.py:126-149).
// ------------------------------------------------------------------------------------------------
#ifndef DDH_BWD_DEEP_PD
#define DDH_BWD_DEEP_PD 2       // register sets of the deep backward sweep: 2 x 40 + the 68 of the window fit 256 registers, 4 do not
#endif
template <int WT, int PD>
__global__ void __launch_bounds__(256, 1)
solve_backward_deep_kernel(PencilDev P, LuDev L, double *__restrict__ xout, const double *__restrict__ pband,
                           const unsigned char *__restrict__ skip) {
    constexpr int NF = 2;
    static_assert(PD >= 2 && PD % 2 == 0, "rows are processed in pairs");
    extern __shared__ int s_lds[];
    const int n = L.n, nb = L.nb, kl = L.kl;
    int *s_perm = s_lds;
    int *s_perm2 = s_lds + (L.pair ? n : 0);
    unsigned char *s_code = (unsigned char *)(s_perm2 + n);
    unsigned char *s_skip = s_code + n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        s_perm[i] = L.colperm[i];
        if (L.pair) s_perm2[i] = L.colperm2[i];
        s_code[i] = L.col_code[i];
        s_skip[i] = skip ? (skip[L.colperm[i]] && (!L.pair || skip[L.colperm2[i]])) : 0;
    }
    __syncthreads();
    const int blk = (L.nsplit > 1) ? __builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * blockDim.x) / L.Gp)) : 0;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x - (long)blk * L.Gp;
    const int row0 = (L.nsplit > 1) ? blk * L.nh : 0, row1 = (L.nsplit > 1) ? row0 + L.nh : n;
    const SysId id = sys_id<true>(P, L, g);
    if (!id.ok) return;
    const int s = id.s;
    const CellCtx c = cell_ctx(P, id.cell);
    const long G = id.G;
    const int *my_perm = id.partner ? s_perm2 : s_perm;
    const bool conjq = id.partner && s == 1;
    const long plane = P.nx * P.ny;

    double2 win[WT];   // win[d] = x[j+1+d] (graded)
#pragma unroll
    for (int d = 0; d < WT; ++d) {
        win[d] = make_double2(0.0, 0.0);
        if (d < nb && row1 == n) win[d] = L.scratch[(long)(n + d) * G + g];
    }
    constexpr int NQ = (WT + 2) / 2;
    // the loaded words stay as the load instructions return them (four dwords): any conversion at the request would be an
    // instruction that waits for the data right there
    ddh_u4v u[PD][NQ];
    double pr[PD][PBW];
    ddh_u4v yv[PD];
    const double *const ur0 = (const double *)L.Aw + lu_aw(L, id.gl, 0, kl);
    const double2 *const y0 = L.scratch + g;
    const __amdgpu_buffer_rsrc_t ur_rs = wave_rsrc(ur0);
    const unsigned ur_lane = wave_lane_off(ur0);
    const __amdgpu_buffer_rsrc_t y_rs = wave_rsrc(y0);
    const unsigned y_lane = wave_lane_off(y0);
    const unsigned ur_row8 = (unsigned)L.BW << 9, y_row16 = (unsigned)(G * (long)sizeof(double2));
    auto lo = [](const ddh_u4v &q) -> double { return __hiloint2double((int)q.y, (int)q.x); };
    auto hi = [](const ddh_u4v &q) -> double { return __hiloint2double((int)q.w, (int)q.z); };
    auto fetch = [&](int j, ddh_u4v *uu, ddh_u4v &y, double *p) {
        y = __builtin_amdgcn_raw_buffer_load_b128(y_rs, y_lane, (unsigned)j * y_row16, 0);
        const double *prow = pband + (long)j * PBW;
#pragma unroll
        for (int d = 0; d < PBW; ++d) p[d] = prow[d];
        const unsigned urow = (unsigned)j * ur_row8;
#pragma unroll
        for (int q = 0; q < NQ; ++q) uu[q] = __builtin_amdgcn_raw_buffer_load_b128(ur_rs, ur_lane, urow + (unsigned)(q << 10), 0);
        // the requests stay HERE: left alone, the scheduler sinks them to their first use (fewer live registers) and the
        // sweep is a chain of exposed round trips again
        __builtin_amdgcn_sched_barrier(0);
    };
    auto emit = [&](int j, double2 v) {
        if (s_skip[j]) return;                           // wave-uniform
        const unsigned char code = s_code[j];
        if (code & 1) v = make_double2(-v.y, v.x);
        if ((code & 2) && s == 1) v = make_double2(-v.x, -v.y);
        if (conjq) v.y = -v.y;
        store_sys<NF>(xout, plane, my_perm[j], P, c, s, v);
    };
    auto ent = [&](const ddh_u4v *uu, int e) -> double { return (e & 1) ? hi(uu[e >> 1]) : lo(uu[e >> 1]); };
    auto row_even = [&](int j, const ddh_u4v *uu, const ddh_u4v &yr, const double *p) -> double2 {
        double2 acc = make_double2(lo(yr), hi(yr));
#pragma unroll
        for (int d = 0; d < WT; ++d) El<true>::fms2(acc, ent(uu, d + 1), win[d]);
        const double2 xj = El<true>::mul2(acc, ent(uu, 0));
        double2 v = xj;
#pragma unroll
        for (int d = 0; d < PBW; ++d)
            if (d < WT) { v.x += p[d] * win[d].x; v.y += p[d] * win[d].y; }
        emit(j, v);
        return xj;
    };
    auto row_odd = [&](int j, const ddh_u4v *uu, const ddh_u4v &yr, double2 xprev, const double *p) {
        double2 acc = make_double2(lo(yr), hi(yr));
        El<true>::fms2(acc, ent(uu, 1), xprev);
#pragma unroll
        for (int d = 1; d < WT; ++d) El<true>::fms2(acc, ent(uu, d + 1), win[d - 1]);
        const double2 xj = El<true>::mul2(acc, ent(uu, 0));
        double2 v = xj;
        v.x += p[0] * xprev.x;
        v.y += p[0] * xprev.y;
#pragma unroll
        for (int d = 1; d < PBW; ++d)
            if (d - 1 < WT) { v.x += p[d] * win[d - 1].x; v.y += p[d] * win[d - 1].y; }
        emit(j, v);
#pragma unroll
        for (int d = WT - 1; d > 1; --d) win[d] = win[d - 2];
        win[1] = xprev;
        win[0] = xj;
    };
    int j = row1 - 1;
    // row (row1 - 1 - k) travels in register set k % PD
#pragma unroll
    for (int d = 0; d < PD; ++d) fetch(max(j - d, row0), u[d], yv[d], pr[d]);
    for (; j - (PD - 1) >= row0; j -= PD) {
#pragma unroll
        for (int d = 0; d < PD; d += 2) {
            // (the requests are unconditional -- past the end the first row again: with a branch around them the number of
            //  loads in flight would depend on the path and the compiler would wait for the smaller count, i.e. for all)
            const double2 xe = row_even(j - d, u[d], yv[d], pr[d]);
            fetch(max(j - d - PD, row0), u[d], yv[d], pr[d]);
            row_odd(j - d - 1, u[d + 1], yv[d + 1], xe, pr[d + 1]);
            fetch(max(j - d - 1 - PD, row0), u[d + 1], yv[d + 1], pr[d + 1]);
        }
    }
    // fewer than PD rows are left, requested into sets 0, 1, ... in order
#pragma unroll
    for (int d = 0; d < PD; d += 2) {
        if (j - d - 1 >= row0) {
            const double2 xe = row_even(j - d, u[d], yv[d], pr[d]);
            row_odd(j - d - 1, u[d + 1], yv[d + 1], xe, pr[d + 1]);
        } else if (j - d >= row0) {
            row_even(j - d, u[d], yv[d], pr[d]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward sweep of the real-graded two-axis path with the factor rows and the forward sweep's vector y staged through a
// per-wave LDS RING filled by LDS-DMA (buffer_load_dwordx4 ... lds), D rows ahead.
//
// solve_backward_kernel<2, 17, true, false, true, 0, 4> (the kernel this one replaces at 512^2 pencils) has no prefetch at
// all: its 128 registers hold the solution window, and a row is "request 9 + 1 loads, wait one HBM round trip, 40 FMAs,
// store" with only the four waves of a SIMD to hide each other's round trips -- 3.6 TB/s, 0.42 of the wave cycles waiting
// on memory.  A register prefetch costs the fourth wave (tried: slower).  LDS-DMA loads have NO destination registers, so
// the rows of the next D steps can be in flight whatever the register file holds:
//   * the four lanes that share a factorization (P, Q systems of a cell and of its transposed partner: sys_id) used to
//     request the same 16 bytes four times; here the wave fetches each row ONCE: lane l of DMA instruction i brings entry
//     pair q = 4 i + l / 16 of the factorization of lane group l % 16 -- 9 x 256 B per row in three instructions -- and every
//     lane reads its pairs back with broadcast ds_read_b128 (four lanes per address: conflict free);
//   * y (64 x 16 contiguous bytes per row) comes by one more DMA instruction;
//   * the loads are hand-issued and waited for with s_waitcnt vmcnt(NV (D - 1)): loads retire in order among themselves,
//     so "at most the NV (D - 1) operations of the D - 1 younger rows outstanding" means row j has landed whatever the
//     stores of x in between do (csrc/ddh_gridwave2.hip uses the same argument); past the end of the sweep the wave keeps
//     requesting its last row so that the count stays uniform.
// Same arithmetic in the same order as solve_backward_kernel: results are bit-identical (tests/test_gpu_pencil.py).
// Waves whose lane quads do not share factorizations (the unpaired cells of the axes and the diagonal, a ragged last
// wave) take the direct-load loop of the old kernel.
// Replaces the back substitution of the reference's per-pencil SuperLU solve (libraries/matsolvers.py:126-149,
// core/timesteppers.py:630-643).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void *ddh_ldsptr;

__device__ __forceinline__ void ring_dma16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned lds_dst, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rs), "s"(lds_dst), "s"(soff)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void ring_wait() {
    static_assert(N >= 0 && N <= 32, "vmcnt immediate");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else static_assert(N % 4 == 0 && N <= 20, "instantiated depths");
}

template <int WT>
struct BwdRing {
    static constexpr int NQ = (WT + 2) / 2;           // entry pairs of a row from the diagonal on (WT + 1 entries)
    static constexpr int NI = (NQ + 3) / 4;           // DMA instructions per factor row (four pairs x 16 factorizations each)
    static constexpr int FB = NQ * 256;               // factor bytes per row and wave
    static constexpr int SLOT = FB + 1024;            // + y
    static constexpr int NV = NI + 1;                 // vector-memory operations per row
};

template <int WT, int D, int MINW>
__global__ void __launch_bounds__(256, MINW)
solve_backward_ring_kernel(PencilDev P, LuDev L, double *__restrict__ xout, const double *__restrict__ pband,
                           const unsigned char *__restrict__ skip) {
    constexpr int NF = 2;
    using RG = BwdRing<WT>;
    static_assert(RG::NV == 4, "ring_wait immediates are multiples of four");
    static_assert((WT & 1) == 1, "an even number of entries per row: WT + 1");
    extern __shared__ int s_lds[];
    const int n = L.n, nb = L.nb, kl = L.kl;
    int *s_perm = s_lds;
    int *s_perm2 = s_lds + (L.pair ? n : 0);
    unsigned char *s_code = (unsigned char *)(s_perm2 + n);
    unsigned char *s_skip = s_code + n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        s_perm[i] = L.colperm[i];
        if (L.pair) s_perm2[i] = L.colperm2[i];
        s_code[i] = L.col_code[i];
        s_skip[i] = skip ? (skip[L.colperm[i]] && (!L.pair || skip[L.colperm2[i]])) : 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // this wave's ring: behind the tables, 1 KiB aligned relative to the start of the dynamic LDS
    const unsigned tab_bytes = (unsigned)(((size_t)n * (L.pair ? 10 : 6) + 1023) & ~(size_t)1023);
    char *ring = reinterpret_cast<char *>(s_lds) + tab_bytes + (size_t)wave * (D * RG::SLOT);
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(ddh_ldsptr)ring);

    const int blk = (L.nsplit > 1) ? __builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * blockDim.x) / L.Gp)) : 0;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x - (long)blk * L.Gp;
    const int row0 = (L.nsplit > 1) ? blk * L.nh : 0, row1 = (L.nsplit > 1) ? row0 + L.nh : n;
    const SysId id = sys_id<true>(P, L, g);
    if (!ring_wave(id, lane)) return;                // (solve_backward_kernel<..., 64> sweeps this wave's systems)
    const int s = id.s;
    const CellCtx c = cell_ctx(P, id.cell);
    const long G = id.G;
    const int *my_perm = id.partner ? s_perm2 : s_perm;
    const bool conjq = id.partner && s == 1;
    const long plane = P.nx * P.ny;

    double2 win[WT];   // win[d] = x[j+1+d] (graded)
#pragma unroll
    for (int d = 0; d < WT; ++d) {
        win[d] = make_double2(0.0, 0.0);
        if (d < nb && row1 == n) win[d] = L.scratch[(long)(n + d) * G + g];
    }
    unsigned dma_voff, y_lane, ur_row8, y_row16;
    __amdgpu_buffer_rsrc_t ur_rs, y_rs;
    {
        const double *const ur0 = (const double *)L.Aw + lu_aw(L, id.gl, 0, kl);     // row 0, diagonal
        const double2 *const y0 = L.scratch + g;
        ur_rs = wave_rsrc(ur0);
        y_rs = wave_rsrc(y0);
        y_lane = wave_lane_off(y0);
        ur_row8 = (unsigned)L.BW << 9;
        y_row16 = (unsigned)(G * (long)sizeof(double2));
        // DMA role of this lane: entry pair (lane >> 4) (+ 4 per instruction) of the factorization of lane quad (lane & 15)
        dma_voff = (unsigned)__shfl((int)wave_lane_off(ur0), 4 * (lane & 15)) + ((unsigned)(lane >> 4) << 10);
    }
    // the ring is read through LDS addresses of this lane: pair q of its factorization at q * 256 + (lane / 4) * 16, y behind
    const unsigned rd_u = (unsigned)(lane >> 2) * 16u, rd_y = (unsigned)RG::FB + (unsigned)lane * 16u;

    auto emit = [&](int j, double2 v) {
        if (s_skip[j]) return;                           // wave-uniform
        const unsigned char code = s_code[j];
        if (code & 1) v = make_double2(-v.y, v.x);
        if ((code & 2) && s == 1) v = make_double2(-v.x, -v.y);
        if (conjq) v.y = -v.y;
        store_sys<NF>(xout, plane, my_perm[j], P, c, s, v);
    };
    unsigned slot_issue = 0;                         // byte offset of the slot the next request fills (uniform)
    auto issue = [&](int j) {
        const int jj = j < row0 ? row0 : j;          // past the end: the last row again (keeps the operation count uniform)
        const unsigned dst = ring_lds + slot_issue;
        const unsigned urow = (unsigned)jj * ur_row8;
#pragma unroll
        for (int i = 0; i < RG::NI; ++i) {
            if (4 * (i + 1) <= RG::NQ) {
                ring_dma16(dma_voff, ur_rs, dst + 1024u * i, urow + 4096u * i);
            } else if (4 * i + (lane >> 4) < RG::NQ) {           // the last instruction brings NQ - 4 i pairs
                ring_dma16(dma_voff, ur_rs, dst + 1024u * i, urow + 4096u * i);
            }
        }
        ring_dma16(y_lane, y_rs, dst + RG::FB, (unsigned)jj * y_row16);
        slot_issue += RG::SLOT;
        if (slot_issue == (unsigned)(D * RG::SLOT)) slot_issue = 0;
    };
    unsigned slot_read = 0;
    // One row: x_j = (y_j - sum_d U[j, j + d] x_(j + d)) / U[j, j], emitted as x_j + sum_d P[j, j + d] x_(j + d).
    // ODD: the row above (j + 1) was solved in this pair and sits in `xprev`, the window still starts at j + 2.
    // The entry pairs are read from the ring three at a time, between the multiply-adds that consume them (the window
    // fills the register file: 36 more registers for a whole row would spill).
    auto row = [&](int j, auto odd_tag, double2 xprev) -> double2 {
        constexpr bool ODD = decltype(odd_tag)::value;
        ring_wait<RG::NV * (D - 1)>();               // row j has landed (the D - 1 younger requests may be in flight)
        const char *rs = ring + slot_read;
        double2 acc = *reinterpret_cast<const double2 *>(rs + rd_y);
        double pr[PBW];
        {
            const double *prow = pband + (long)j * PBW;
#pragma unroll
            for (int d = 0; d < PBW; ++d) pr[d] = prow[d];
        }
        double piv = 0.0;
#pragma unroll
        for (int q = 0; q < RG::NQ; ++q) {
            const double2 uu = *reinterpret_cast<const double2 *>(rs + rd_u + 256 * q);
            // entries 2 q, 2 q + 1 of the row: entry e multiplies x_(j + e)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * q + h;
                const double ue = h ? uu.y : uu.x;
                if (e == 0) {
                    piv = ue;                            // reciprocal pivot
                } else if (e <= WT) {
                    if (ODD) {
                        if (e == 1) El<true>::fms2(acc, ue, xprev);
                        else El<true>::fms2(acc, ue, win[e - 2]);
                    } else {
                        El<true>::fms2(acc, ue, win[e - 1]);
                    }
                }
            }
            if (q % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
        const double2 xj = El<true>::mul2(acc, piv);
        // the slot has been read by every lane: request row j - D into it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue(j - D);
        slot_read += RG::SLOT;
        if (slot_read == (unsigned)(D * RG::SLOT)) slot_read = 0;
        double2 v = xj;
        if (ODD) {
            v.x += pr[0] * xprev.x;
            v.y += pr[0] * xprev.y;
#pragma unroll
            for (int d = 1; d < PBW; ++d)
                if (d - 1 < WT) { v.x += pr[d] * win[d - 1].x; v.y += pr[d] * win[d - 1].y; }
        } else {
#pragma unroll
            for (int d = 0; d < PBW; ++d)
                if (d < WT) { v.x += pr[d] * win[d].x; v.y += pr[d] * win[d].y; }
        }
        emit(j, v);
        return xj;
    };
    int j = row1 - 1;
#pragma unroll
    for (int d = 0; d < D; ++d) issue(j - d);
    while (j >= row0 + 1) {
        const double2 xe = row(j, std::false_type(), make_double2(0.0, 0.0));
        const double2 xo = row(j - 1, std::true_type(), xe);
#pragma unroll
        for (int d = WT - 1; d > 1; --d) win[d] = win[d - 2];
        win[1] = xe;
        win[0] = xo;
        j -= 2;
    }
    if (j == row0) row(row0, std::false_type(), make_double2(0.0, 0.0));
    ring_wait<0>();                                  // nothing of this wave may land in LDS after it has gone
}

// Border unknowns of a forward sweep that ran one thread per (system, block) (LuDev::nsplit > 1): the blocks' partial
// border sums (scratch rows n + nb ...) are added, the Schur block applied, the border unknowns stored -- the tail of
// solve_forward_lean_kernel, one thread per system.
template <int NF>
__global__ void __launch_bounds__(256)
border_finish_kernel(PencilDev P, LuDev L, double *__restrict__ xout) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const SysId id = sys_id<true>(P, L, g);
    if (!id.ok) return;
    const int s = id.s, n = L.n, nb = L.nb, N = L.N;
    const CellCtx c = cell_ctx(P, id.cell);
    const long G = id.G, plane = P.nx * P.ny;
    const bool conjq = id.partner && s == 1;
    const double *Ab = (const double *)L.Ab;
    double2 gb[NBMAX];
    for (int r = 0; r < nb; ++r) {
        double2 acc = make_double2(0.0, 0.0);
        for (int b = 0; b < L.nsplit; ++b) {
            const double2 v = L.scratch[(long)(n + nb + b * nb + r) * G + g];
            acc.x += v.x;
            acc.y += v.y;
        }
        gb[r] = acc;
    }
    for (int r = 0; r < nb; ++r) {
        double2 acc = make_double2(0.0, 0.0);
        for (int cidx = 0; cidx < nb; ++cidx) El<true>::fma2(acc, Ab[lu_ab(L, id.gl, n + cidx, r)], gb[cidx]);
        L.scratch[(long)(n + r) * G + g] = acc;
        double2 v = acc;
        const unsigned char code = L.col_code[n + r];
        if (code & 1) v = make_double2(-v.y, v.x);
        if ((code & 2) && s == 1) v = make_double2(-v.x, -v.y);
        if (conjq) v.y = -v.y;
        const int prow = id.partner ? L.colperm2[n + r] : L.colperm[n + r];
        store_sys<NF>(xout, plane, prow, P, c, s, v);
    }
    (void)N;
}


// ------------------------------------------------------------------------------------------------
// Cooperative sweeps for FEW systems (2-D problems, small 3-D ones): with one thread per system a launch of a few
// hundred systems leaves the chip empty and every row costs a full memory latency (2-D Rayleigh-Benard 512 x 256:
// 0.6-0.8 us per row, 2060 rows).  Here CH = 16 lanes share one system:
//  * forward: lane h holds the window row i = h (mod 16) (kl < 16: one live row per lane); per step the pivot row
//    and row j are exchanged by two lane broadcasts, every lane updates its own row with its own multiplier;
//  * backward: lane h holds the solution entries x[k], k = h (mod 16), inside the upper band; per row each lane
//    forms the partial dot product over its <= ceil(W/16) entries, a 4-step butterfly adds the partials;
//  * every lane prefetches its own few factor entries COOP_D rows ahead (a rotating register file indexed at
//    compile time), so the per-row cost is a handful of FMAs and lane exchanges instead of a memory round trip.
// Results are identical in exact arithmetic and equal up to the summation order in floating point.
// ------------------------------------------------------------------------------------------------
constexpr int CH = 16;
constexpr int COOP_D = 8;

// lane exchange inside a row of 16 lanes by DPP (VALU speed; a ds_bpermute round trip costs ~100 cycles and the
// backward sweep has four dependent ones per row)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// sum over aligned groups of CB = 4, 8 or 16 lanes, identical bits in every lane of the group (each step adds a lane
// and its mirror partner)
template <int CB>
__device__ __forceinline__ double group_sum(double v) {
    static_assert(CB == 4 || CB == 8 || CB == 16, "DPP groups of 4, 8 or 16 lanes");
    v += dpp_f64<0xB1>(v);                   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);                   // quad_perm [2,3,0,1]
    if (CB >= 8) v += dpp_f64<0x141>(v);     // row_half_mirror
    if (CB >= 16) v += dpp_f64<0x140>(v);    // row_mirror
    return v;
}

template <int NF, bool REAL, int NBT>
__global__ void __launch_bounds__(256)
solve_forward_coop_kernel(PencilDev P, LuDev L, const double *__restrict__ rhs, double *__restrict__ xout) {
    typedef typename El<REAL>::T E;
    extern __shared__ int s_lds[];
    const int N = L.N;
    int *s_perm = s_lds;
    unsigned char *s_code = (unsigned char *)(s_lds + N + L.nb);
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        s_perm[i] = L.rowperm[i];
        s_code[i] = REAL ? L.row_code[i] : 0;
    }
    for (int i = threadIdx.x; i < L.nb; i += blockDim.x) {
        s_perm[N + i] = L.colperm[L.n + i];
        s_code[N + i] = REAL ? L.col_code[L.n + i] : 0;
    }
    __syncthreads();
    const int h = threadIdx.x & (CH - 1);
    // independent diagonal blocks (LuDev::nsplit, real factors): this lane group sweeps rows row0 .. row1 - 1 of system g
    const int blk = (REAL && L.nsplit > 1) ? __builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * (256 / CH)) / L.Gp)) : 0;
    const long g = (long)blockIdx.x * (256 / CH) + (threadIdx.x / CH) - (long)blk * L.Gp;
    if (g >= P.G) return;   // whole lane groups leave together; P/Q partner groups are adjacent and leave together
    const int row0 = (REAL && L.nsplit > 1) ? blk * L.nh : 0, row1 = (REAL && L.nsplit > 1) ? row0 + L.nh : L.n;
    const long cell = g / P.S;
    const int s = (int)(g % P.S);
    const CellCtx c = cell_ctx(P, cell);
    const long G = P.G;
    const long gl = REAL ? cell : g;
    const E *Aw = (const E *)L.Aw, *Ab = (const E *)L.Ab;
    const long plane = P.nx * P.ny;
    const int n = L.n, nb = L.nb, kl = L.kl;

    auto load_row = [&](int i) -> double2 {
        double2 v = load_sys<NF, CH>(rhs, plane, s_perm[i], P, c, s);
        if (REAL) {
            const unsigned char code = s_code[i];
            if ((code & 2) && s == 1) v = make_double2(-v.x, -v.y);
            if (code & 1) v = make_double2(v.y, -v.x);
        }
        return v;
    };

    double2 wv = make_double2(0.0, 0.0);          // window row i = h (mod CH), i in [j, j + kl]
    {
        const int i0 = row0 + ((h - row0) & (CH - 1));
        if (i0 - row0 <= kl && i0 < row1) wv = load_row(i0);
    }
    double2 gb[NBT];
#pragma unroll
    for (int rb = 0; rb < NBT; ++rb) {
        gb[rb] = make_double2(0.0, 0.0);
        if (rb < nb && blk == 0) gb[rb] = load_row(n + rb);       // (block 0 carries the border's right-hand side)
    }
    int pp[COOP_D];
    E pm[COOP_D], pab[COOP_D][NBT];
    double2 pr[COOP_D];
    // branch-free: every lane issues the same number of loads per row (clamped addresses, masked values), so the
    // compiler can wait for exactly the oldest outstanding row (s_waitcnt vmcnt(k)) instead of draining the queue
    // issue() is called for jj = 0, 1, 2, ...: running pointers instead of 64-bit index arithmetic per entry (the
    // address computations were the dominant VALU work of a row)
    const long aw_rs = (long)L.BW * 64, ab_rs = (long)L.nb * 64, pv_rs = 64;
    const int aw_step = (int)(aw_rs - 64);                   // (row + 1, d - 1) relative to (row, d) ...
    const int pk63 = L.pk63;                                 // ... minus 63 when d - 1 is the second member of a pair
    const unsigned char *pv_ptr = L.piv + lu_pv(L, gl, row0);
    const E *aw_ptr = Aw + lu_aw(L, gl, row0, kl);           // (row jj, diagonal)
    const E *ab_ptr = Ab + lu_ab(L, gl, row0, 0);
    auto issue = [&](int jj, int slot) {
        const int dl = (h - jj) & (CH - 1);
        const bool live = dl >= 1 && dl <= kl && jj + dl < row1;
        const int dc = live ? dl : 0;                       // the diagonal itself is always a valid entry
        pp[slot] = *pv_ptr;
        // raw values only: whether an entry is used is decided when the row is consumed (touching the value here
        // would make the wave wait for the load it has just issued)
        pm[slot] = aw_ptr[dc * aw_step - pk63 * (dc & 1)];    // (row jj + dc, column jj); the diagonal's raw entry is even
#pragma unroll
        for (int rb = 0; rb < NBT; ++rb) pab[slot][rb] = ab_ptr[(rb < nb ? rb : 0) << 6];
        const int nxt = jj + kl + 1;
        pr[slot] = load_row(nxt < row1 ? nxt : row1 - 1);    // every lane of the group reads the same row
        const bool adv = jj < row1 - 1;                      // past the end the last row is re-read (and never used)
        pv_ptr += adv ? pv_rs : 0;
        aw_ptr += adv ? aw_rs : 0;
        ab_ptr += adv ? ab_rs : 0;
    };
    double2 *sc_ptr = L.scratch + (long)row0 * G + g;
#pragma unroll
    for (int r = 0; r < COOP_D; ++r) issue(row0 + r, r);
    // One row.  The main loop runs whole blocks of COOP_D rows without any guard: straight-line code lets the compiler
    // wait for exactly the oldest outstanding prefetch (vmcnt(k)); a guard per row merges control-flow paths with
    // different numbers of issued loads and degrades every wait to "almost everything".
#define DDH_COOP_FWD_ROW(r, j)                                                                                     \
    {                                                                                                              \
        const int p = pp[r];                                                                                       \
        const E m = pm[r];                                                                                         \
        E ab[NBT];                                                                                                 \
        _Pragma("unroll") for (int rb = 0; rb < NBT; ++rb) ab[rb] = pab[r][rb];                                    \
        const double2 rnew = pr[r];                                                                                \
        issue((j) + COOP_D, r);                                                                                    \
        const int a = (j) & (CH - 1), bq = ((j) + p) & (CH - 1);                                                   \
        double2 yj, wj;                                                                                            \
        yj.x = __shfl(wv.x, bq, CH);                                                                               \
        yj.y = __shfl(wv.y, bq, CH);                                                                               \
        wj.x = __shfl(wv.x, a, CH);                                                                                \
        wj.y = __shfl(wv.y, a, CH);                                                                                \
        if (h == bq) wv = wj; /* the interchange (a no-op for p = 0) */                                            \
        *sc_ptr = yj; /* row j of the scratch (all lanes of the group store the same value) */                     \
        sc_ptr += G;                                                                                               \
        const int dl = (h - (j)) & (CH - 1);                                                                       \
        if (dl >= 1 && dl <= kl && (j) + dl < row1) El<REAL>::fms2(wv, m, yj);                                     \
        _Pragma("unroll") for (int rb = 0; rb < NBT; ++rb) if (rb < nb) El<REAL>::fms2(gb[rb], ab[rb], yj);        \
        const int nxt = (j) + kl + 1;                                                                              \
        if (nxt < row1 && (nxt & (CH - 1)) == h) wv = rnew;                                                        \
    }
    int j0 = row0;
    for (; j0 + COOP_D <= row1; j0 += COOP_D) {
#pragma unroll
        for (int r = 0; r < COOP_D; ++r) DDH_COOP_FWD_ROW(r, j0 + r)
    }
#pragma unroll
    for (int r = 0; r < COOP_D; ++r)
        if (j0 + r < row1) DDH_COOP_FWD_ROW(r, j0 + r)
#undef DDH_COOP_FWD_ROW
    if (REAL && L.nsplit > 1) {
        // the border rows collect contributions of every block: partial sums, finished by border_finish_kernel
        if (h == 0) {
#pragma unroll
            for (int r = 0; r < NBT; ++r)
                if (r < nb) L.scratch[(long)(n + nb + blk * nb + r) * G + g] = gb[r];
        }
        return;
    }
    // ---- Schur block (every lane of the group computes it; lane 0 stores)
#pragma unroll
    for (int r = 0; r < NBT; ++r) {
        if (r < nb) {
            double2 acc = make_double2(0.0, 0.0);
#pragma unroll
            for (int cidx = 0; cidx < NBT; ++cidx)
                if (cidx < nb) El<REAL>::fma2(acc, Ab[lu_ab(L, gl, n + cidx, r)], gb[cidx]);
            L.scratch[(long)(n + r) * G + g] = acc;
            double2 v = acc;
            if (REAL) {
                const unsigned char code = s_code[N + r];
                if (code & 1) v = make_double2(-v.y, v.x);
                if ((code & 2) && s == 1) v = make_double2(-v.x, -v.y);
            }
            store_sys<NF, CH>(xout, plane, s_perm[N + r], P, c, s, v);
        }
    }
}

template <int NF, int TT, bool REAL, int CB>
__global__ void __launch_bounds__(256)
solve_backward_coop_kernel(PencilDev P, LuDev L, double *__restrict__ xout) {
    typedef typename El<REAL>::T E;
    // rows prefetched ahead: 8 for the 16-lane variant (2-3 entries per lane and row); the 4-lane variant holds 9-12
    // entries per lane and row -- 8 rows of them are 144+ registers and spilled 420-3000 bytes per lane (one wave per SIMD)
    constexpr int PD = (CB <= 4) ? 4 : COOP_D;
    extern __shared__ int s_lds[];
    const int n = L.n, nb = L.nb, kl = L.kl, W = L.W;
    int *s_perm = s_lds;
    unsigned char *s_code = (unsigned char *)(s_lds + n);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        s_perm[i] = L.colperm[i];
        s_code[i] = REAL ? L.col_code[i] : 0;
    }
    __syncthreads();
    const int h = threadIdx.x & (CB - 1);
    // independent diagonal blocks (LuDev::nsplit, real factors): rows row1 - 1 .. row0 of system g
    const int blk = (REAL && L.nsplit > 1) ? __builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * (256 / CB)) / L.Gp)) : 0;
    const long g = (long)blockIdx.x * (256 / CB) + (threadIdx.x / CB) - (long)blk * L.Gp;
    if (g >= P.G) return;
    const int row0 = (REAL && L.nsplit > 1) ? blk * L.nh : 0, row1 = (REAL && L.nsplit > 1) ? row0 + L.nh : n;
    const long cell = g / P.S;
    const int s = (int)(g % P.S);
    const CellCtx c = cell_ctx(P, cell);
    const long G = P.G;
    const long gl = REAL ? cell : g;
    const E *Aw = (const E *)L.Aw;
    const long plane = P.nx * P.ny;

    // xr[t] = x[k_t]: the entries k = h (mod CB) above the current row, k_t = j + 1 + e + CB t, e = (h - j - 1) mod CB
    double2 xr[TT];
    {
        const int e = (h - row1) & (CB - 1);           // row j = row1 - 1
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int kb = e + CB * t;                 // k - row1 (border columns follow the LAST band rows)
            xr[t] = (kb < nb && row1 == n) ? L.scratch[(long)(n + kb) * G + g] : make_double2(0.0, 0.0);
        }
    }
    E pu[PD][TT], pu0[PD];
    double2 py[PD];
    // issue() is called for jj = n - 1, n - 2, ...: running pointers (see the forward kernel)
    const long aw_rs = (long)L.BW * 64;
    const int pk63 = L.pk63;
    const E *u_ptr = Aw + lu_aw(L, gl, row1 - 1, kl);
    const double2 *y_ptr = L.scratch + (long)(row1 - 1) * G + g;
    auto issue = [&](int jj, int slot) {                  // branch-free
        py[slot] = *y_ptr;
        pu0[slot] = u_ptr[0];
        const int e = (h - jj - 1) & (CB - 1);
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int d = 1 + e + CB * t;
            const int dd = (d <= W) ? d : 0;
            pu[slot][t] = u_ptr[(dd << 6) - pk63 * (dd & 1)];       // raw; entries beyond the band are dropped at use
        }
        const bool adv = jj > row0;
        u_ptr -= adv ? aw_rs : 0;
        y_ptr -= adv ? G : 0;
    };
#pragma unroll
    for (int r = 0; r < PD; ++r) issue(row1 - 1 - r, r);
#define DDH_COOP_BWD_ROW(r, j)                                                                                     \
    {                                                                                                              \
        E u[TT];                                                                                                   \
        const int ej = (h - (j) - 1) & (CB - 1);                                                                   \
        _Pragma("unroll") for (int t = 0; t < TT; ++t) u[t] = (1 + ej + CB * t <= W) ? pu[r][t] : El<REAL>::zero(); \
        const E u0 = pu0[r];                                                                                       \
        const double2 y = py[r];                                                                                   \
        issue((j) - PD, r);                                                                                    \
        double2 acc = make_double2(0.0, 0.0);                                                                      \
        _Pragma("unroll") for (int t = 0; t < TT; ++t) El<REAL>::fma2(acc, u[t], xr[t]);                           \
        acc.x = group_sum<CB>(acc.x);                                                                                  \
        acc.y = group_sum<CB>(acc.y);                                                                                  \
        const double2 xj = El<REAL>::mul2(make_double2(y.x - acc.x, y.y - acc.y), u0);                             \
        double2 v = xj;                                                                                            \
        if (REAL) {                                                                                                \
            const unsigned char code = s_code[j];                                                                  \
            if (code & 1) v = make_double2(-v.y, v.x);                                                             \
            if ((code & 2) && s == 1) v = make_double2(-v.x, -v.y);                                                \
        }                                                                                                          \
        store_sys<NF, CB>(xout, plane, s_perm[j], P, c, s, v); /* (same value from all lanes of the group) */      \
        if ((((j) - h) & (CB - 1)) == 0) { /* the lane that owns k = j takes the new entry */                      \
            _Pragma("unroll") for (int t = TT - 1; t > 0; --t) xr[t] = xr[t - 1];                                  \
            xr[0] = xj;                                                                                            \
        }                                                                                                          \
    }
    int jt0 = 0;
    const int nrows = row1 - row0;
    for (; jt0 + PD <= nrows; jt0 += PD) {       // guard-free whole blocks (see the forward kernel)
#pragma unroll
        for (int r = 0; r < PD; ++r) DDH_COOP_BWD_ROW(r, row1 - 1 - (jt0 + r))
    }
#pragma unroll
    for (int r = 0; r < PD; ++r)
        if (jt0 + r < nrows) DDH_COOP_BWD_ROW(r, row1 - 1 - (jt0 + r))
#undef DDH_COOP_BWD_ROW
}

// ------------------------------------------------------------------------------------------------
// dense fallback for flagged cells: x = Inv * rhs (explicit inverse built on the host)
// ------------------------------------------------------------------------------------------------
template <int NF>
__global__ void __launch_bounds__(256)
dense_gather_kernel(PencilDev P, LuDev L, const long *__restrict__ cells, int ncellsf,
                    const RhsSrc rhs, double2 *__restrict__ out) {
    // thread -> (flagged cell f, system s, logical row i); pairs (s=0,1) adjacent lanes for NF == 2
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)L.N * P.S;
    if (t >= per * ncellsf) return;
    const long f = t / per;
    const int i = (int)((t % per) / P.S);
    const int s = (int)(t % P.S);
    const CellCtx c = cell_ctx(P, cells[f]);
    const double2 v = load_sys<NF>(rhs, P.nx * P.ny, L.rowperm[i], P, c, s);
    out[(f * P.S + s) * L.N + i] = v;
}

__global__ void __launch_bounds__(256)
dense_apply_kernel(int N, const double2 *__restrict__ inv, const double2 *__restrict__ rhs, double2 *__restrict__ x) {
    // one wave per output row: blockIdx.y = system, 4 rows per block
    const int sys = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const double2 *a = inv + ((long)sys * N + row) * N;
    const double2 *b = rhs + (long)sys * N;
    double2 acc = make_double2(0.0, 0.0);
    for (int k = lane; k < N; k += 64) cfma(acc, a[k], b[k]);
    for (int off = 32; off > 0; off >>= 1) {
        acc.x += __shfl_down(acc.x, off);
        acc.y += __shfl_down(acc.y, off);
    }
    if (lane == 0) x[(long)sys * N + row] = acc;
}

template <int NF>
__global__ void __launch_bounds__(256)
dense_scatter_kernel(PencilDev P, LuDev L, const long *__restrict__ cells, int ncellsf,
                     const double2 *__restrict__ xin, double *__restrict__ xout, int apply_p) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)L.N * P.S;
    if (t >= per * ncellsf) return;
    const long f = t / per;
    const int i = (int)((t % per) / P.S);
    const int s = (int)(t % P.S);
    const CellCtx c = cell_ctx(P, cells[f]);
    const double2 *xs = xin + (f * P.S + s) * L.N;
    double2 v = xs[i];
    if (apply_p && L.pband && i < L.n) {
        // (the dense inverse is the plain complex inverse in logical ordering: P is the same real band there)
        const double *pr = L.pband + (long)i * PBW;
        for (int d = 1; d <= PBW; ++d)
            if (i + d < L.n) { v.x += pr[d - 1] * xs[i + d].x; v.y += pr[d - 1] * xs[i + d].y; }
    }
    store_sys<NF>(xout, P.nx * P.ny, L.colperm[i], P, c, s, v);
}

template <typename T>
static int upload_vec(void **dptr, const T *src, size_t count) {
    DDH_HIP(hipMalloc(dptr, count * sizeof(T) + 16));
    if (count) DDH_HIP(hipMemcpy(*dptr, src, count * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Few systems (2-D problems: a few hundred pencils of ~1000 rows): a sweep is a chain of n / nblocks dependent rows behind
// too few wavefronts to hide anything (2-D Rayleigh-Benard 512 x 256: 0.39 ms per solve, 0.02 of the HBM rate).  With
// real-graded factors and the band split into independent diagonal blocks, the explicit inverse of a block is small
// (nh^2 doubles: 2 MB at nh = 515) and its application is a streaming GEMV: one workgroup per (pencil, block), thread i
// forms y_i = sum_k Binv[i][k] r_k from the TRANSPOSED inverse (row k contiguous in i: coalesced, the right-hand side
// broadcast from LDS), for the real and the imaginary part of the graded right-hand side at once.  The inverses come from
// unit solves of the band LU (csrc/ddh_ellband.hip), as for the sphere.  Same linear systems as the reference's
// per-subproblem sparse LU (libraries/matsolvers.py:126-149); flagged pencils keep their dense path.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
blockinv_solve_kernel(PencilDev P, LuDev L, const RhsSrc rhs, const double *__restrict__ binv, double *__restrict__ yout) {
    extern __shared__ double2 s_rb[];                        // the graded right-hand side of the block
    const int nh = L.nh, ns = L.nsplit;
    const long cell = blockIdx.x / ns;
    const int blk = (int)(blockIdx.x - cell * ns);
    if (L.flag[cell]) return;                                // (real factors: one stored factorization per cell)
    const CellCtx c = cell_ctx(P, cell);
    const long plane = P.nx * P.ny;
    const int row0 = blk * nh, tid = threadIdx.x, NT = blockDim.x;
    for (int i = tid; i < nh; i += NT) {
        const int phys = L.rowperm[row0 + i];
        double2 v = make_double2(0.0, 0.0);
        if (!(rhs.zrow && rhs.zrow[phys])) v = load_sys<1>(rhs, plane, phys, P, c, 0);
        if (L.row_code[row0 + i] & 1) v = make_double2(v.y, -v.x);
        s_rb[i] = v;
    }
    __syncthreads();
    const double *B = binv + (size_t)blockIdx.x * nh * nh;
    for (int i = tid; i < nh; i += NT) {
        double2 acc = make_double2(0.0, 0.0);
        const double *bi = B + i;
        int k = 0;
        for (; k + 8 <= nh; k += 8) {
            double b[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) b[q] = bi[(size_t)(k + q) * nh];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double2 r = s_rb[k + q];
                acc.x += b[q] * r.x;
                acc.y += b[q] * r.y;
            }
        }
        for (; k < nh; ++k) {
            const double b = bi[(size_t)k * nh];
            const double2 r = s_rb[k];
            acc.x += b * r.x;
            acc.y += b * r.y;
        }
        double2 v = acc;
        if (L.col_code[row0 + i] & 1) v = make_double2(-v.y, v.x);
        store_sys<1>(yout, plane, L.colperm[row0 + i], P, c, 0, v);
    }
    // border unknowns exist for flagged pencils only (checked by the host before it hands the inverses over): zero here
    if (blk == 0)
        for (int r = tid; r < L.nb; r += NT) store_sys<1>(yout, plane, L.colperm[L.n + r], P, c, 0, make_double2(0.0, 0.0));
}

// dense fallback for the flagged pencils (after either sweep variant)
template <int NF>
static int finish_solve(PencilPack *pp, LuFactor *lu, const RhsSrc &rhs, double *x, hipStream_t s, int apply_p) {
    const PencilDev &P = pp->dev;
    const LuDev &d = lu->dev;
    if (lu->nflag) {
        if (!lu->d_inv) return fail("pencil_solve: flagged pencils need ddh_pencil_set_dense_inverse first");
        const int nsys = lu->nflag * P.S;
        double2 *drhs = (double2 *)lu->d_dense_rhs;
        double2 *dx = drhs + (size_t)nsys * d.N;
        const long work = (long)lu->nflag * d.N * P.S;
        const unsigned gb = (unsigned)((work + 255) / 256);
        hipLaunchKernelGGL(dense_gather_kernel<NF>, dim3(gb), dim3(256), 0, s, P, d, (const long *)lu->d_flag_cells,
                           lu->nflag, rhs, drhs);
        hipLaunchKernelGGL(dense_apply_kernel, dim3((unsigned)((d.N + 3) / 4), (unsigned)nsys), dim3(256), 0, s, d.N,
                           (const double2 *)lu->d_inv, (const double2 *)drhs, dx);
        hipLaunchKernelGGL(dense_scatter_kernel<NF>, dim3(gb), dim3(256), 0, s, P, d, (const long *)lu->d_flag_cells,
                           lu->nflag, (const double2 *)dx, x, apply_p);
        DDH_HIP(hipGetLastError());
    }
    return 0;
}

// Sweep variant by the number of systems (measured on MI355X with the 5156-row systems of the 3-D benchmark, solve =
// both sweeps, ms):
//   G = 16384: one thread per system 5.79 | fwd coop + bwd 16 lanes 4.24 | + bwd 8 lanes 3.49 | + bwd 4 lanes 3.28
//   G = 32768: 6.13 | fwd coop 6.5-8.4 | fwd per-thread + bwd 4 lanes 5.75      G = 65536: 6.76 (anything else slower)
// (these are the per-rank sizes of the 512 x 512 x 256 problem on 8 / 4 / 2 GPUs); 2-D problems (a few hundred
// systems) were tuned with 16 lanes in both sweeps.  forward: 16 lanes per system (needs kl < 16); backward: 16 or 4
// lanes per system (fewer lanes = less redundant work per system, more products per lane).
// multipliers per column held by the one-thread-per-system forward kernel for a lower bandwidth kl
static int forward_window(int kl) { return kl <= 6 ? 6 : (kl <= 12 ? 12 : 16); }
// register window (entries above the diagonal) of the one-thread-per-system backward kernel for an upper bandwidth W
static int backward_window(int W) { return W <= 17 ? 17 : (W <= 32 ? 32 : (W <= 34 ? 34 : (W <= 48 ? 48 : 64))); }

template <int NF>
static void choose_variant(const PencilPack *pp, const LuDev &d, int &use_fwd, int &cb) {
    const PencilDev &P = pp->dev;
    const int W = d.W;
    const int coop_mode = pp->coop_mode;
    const bool coop_auto = coop_mode == 1 && NF > 0;
    use_fwd = (coop_mode == 2 || (coop_auto && P.G <= 16384)) ? 1 : 0;
    cb = 0;
    if (coop_mode == 2 || (coop_auto && P.G <= 1024)) cb = 16;
    else if (coop_auto && P.G <= 16384) cb = 4;        // (round 3, register-lean sweeps: at 32 768 systems one thread per
                                                       //  system is faster again, 5.17 vs 5.58 ms -- profiles/r3_strong_scaling_shares.txt)
    // independent diagonal blocks swept by separate threads (LuDev::nsplit): one thread per (system, block) in both sweeps
    // beats the cooperative variants from a few thousand systems on (round 4, profiles/r4_strong_scaling_shares.txt, solve
    // ms per launch at the per-rank shares of 512 x 512 x 256: 16 384 systems: per-thread both 1.97 | per-thread + cb 4:
    // 2.15 | fwd coop + cb 4: 4.36;  32 768: 2.24 | 3.12 | 8.72;  65 536: 3.28 | 5.03 | 17.3)
    if (coop_auto && d.nsplit > 1 && P.G > 4096) use_fwd = cb = 0;
    if (pp->coop_fwd >= 0) use_fwd = pp->coop_fwd;
    if (pp->coop_cb >= 0) cb = pp->coop_cb;
    if (NF == 0 || d.kl >= CH || d.nb > 8) use_fwd = 0;
    if (NF == 0 || (cb != 4 && cb != 16)) cb = 0;
    if (cb && (W + cb - 1) / cb > (cb == 4 ? 12 : 3)) cb = 0;
    if (d.n <= 0) use_fwd = cb = 0;
}

// The lean forward sweep (real-graded factors, two Fourier axes) is instantiated and tested for the windows of the
// Rayleigh-Benard pencils: kl <= 12, a border of <= 2; other shapes take the general kernel (the wider instantiations fit
// two waves per SIMD only partly and have no test yet).
static bool lean_forward_ok(const LuDev &d) {
    static const int no_lean = getenv("DDH_FWD_LEAN") ? !atoi(getenv("DDH_FWD_LEAN")) : 0;
    return d.real && d.n > 0 && !no_lean && d.kpad + d.kl == forward_window(d.kl) &&
           d.rows_aw >= d.n + forward_window(d.kl) && d.kl <= 12 && d.nb <= 2;
}

// Few systems: the one-thread-per-(system, block) sweeps would run at less than two wavefronts per SIMD (MI355X: 1024
// SIMDs): the deep-prefetch variants (solve_*_deep_kernel) take over.  DDH_SWEEP_DEEP=0 / 1 forces the choice.
static bool sweep_deep(const LuDev &d) {
    static const int env = getenv("DDH_SWEEP_DEEP") ? atoi(getenv("DDH_SWEEP_DEEP")) : -1;
    if (!d.real || d.n <= 0) return false;
    if (env >= 0) return env != 0;
    return (long)d.nsplit * d.Gp <= 2L * 1024 * 64;
}

// workgroup size of the deep sweeps: the largest of 256 / 128 / 64 threads that still gives every CU a workgroup (measured at the
// P = 8 share, 32 768 threads: 2.02 / 1.76 / 1.83 ms per step with 256 / 128 / 64; at 65 536 threads 256 wins: 3.04 / 3.38 / 3.47).
// DDH_DEEP_BLOCK forces one.
static unsigned deep_block(long threads) {
    static const int v = getenv("DDH_DEEP_BLOCK") ? atoi(getenv("DDH_DEEP_BLOCK")) : 0;
    if (v == 64 || v == 128 || v == 256) return (unsigned)v;
    if (threads / 256 >= 256) return 256u;
    return threads / 128 >= 256 ? 128u : 64u;
}

// want_p: the caller asks for x = P y (recombination fused into the backward sweep); *did_p tells whether this launch
// could do it (one-thread-per-system backward kernel of the real-graded 2-axis path with a band table on file).
template <int NF>
static int launch_solve(PencilPack *pp, LuFactor *lu, const RhsSrc &rhs, double *x, hipStream_t s, bool want_p = false,
                        bool *did_p = nullptr) {
    const PencilDev &P = pp->dev;
    const LuDev &d = lu->dev;
    if (did_p) *did_p = false;
    const unsigned blocks = (unsigned)((P.G + 255) / 256);
    const int W = d.W;
    const size_t per_entry = d.pair ? 9 : 5;    // one (two when paired) int permutations + a code byte per row
    const size_t lds_f = (size_t)(d.N + d.nb) * (per_entry + 1) + 16,     // (+ the zero-row flags of the lean sweep)
 lds_b = (size_t)(d.n > 0 ? d.n : 1) * (per_entry + 2) + 16;   // (+ the skip flags and the row-fill table of the backward sweep)
    if (lds_f > 64 * 1024) return fail("pencil_solve: system too large for the LDS permutation cache");
    int use_fwd, cb;
    choose_variant<NF>(pp, d, use_fwd, cb);
    if (d.pair) use_fwd = cb = 0;               // partner pencils: one-thread-per-system sweeps only
    if (use_fwd) {
        const unsigned cblocks = (d.real && d.nsplit > 1) ? (unsigned)(((long)d.nsplit * d.Gp) / (256 / CH))
                                                          : (unsigned)((P.G + (256 / CH) - 1) / (256 / CH));
        // the cooperative sweep's deep branch-free prefetch takes ONE right-hand-side vector: a combination is
        // materialised first (few systems: the extra pass is small next to the latency-bound sweeps)
        const double *rhs1 = rhs.p[0];
        if (rhs.n != 1 || rhs.a[0] != 1.0) {
            const long nel = (long)P.nrows * P.nx * P.ny;
            if (!lu->d_rhs_tmp) DDH_HIP(hipMalloc(&lu->d_rhs_tmp, (size_t)nel * sizeof(double) + 16));
            if (int st0 = ddh_lincomb((double *)lu->d_rhs_tmp, rhs.n, rhs.p, rhs.a, nel, (void *)s)) return st0;
            rhs1 = (const double *)lu->d_rhs_tmp;
        }
#define DDH_CFWD(NBTV)                                                                                             \
    {                                                                                                              \
        if (d.real)                                                                                                \
            hipLaunchKernelGGL((solve_forward_coop_kernel<NF, true, NBTV>), dim3(cblocks), dim3(256), lds_f, s, P, d, rhs1, x); \
        else                                                                                                       \
            hipLaunchKernelGGL((solve_forward_coop_kernel<NF, false, NBTV>), dim3(cblocks), dim3(256), lds_f, s, P, d, rhs1, x); \
    }
        if constexpr (NF > 0) {
            if (d.nb <= 2) DDH_CFWD(2) else DDH_CFWD(8)
            if (d.real && d.nsplit > 1 && d.nb > 0)
                hipLaunchKernelGGL(border_finish_kernel<NF>, dim3(blocks), dim3(256), 0, s, P, d, x);
        }
#undef DDH_CFWD
    }
#define DDH_FWD(KLTV, NBTV)                                                                                        \
    {                                                                                                              \
        if (d.real)                                                                                                \
            hipLaunchKernelGGL((solve_forward_kernel<NF, true, KLTV, NBTV>), dim3(blocks), dim3(256), lds_f, s, P, d, rhs, x); \
        else                                                                                                       \
            hipLaunchKernelGGL((solve_forward_kernel<NF, false, KLTV, NBTV>), dim3(blocks), dim3(256), lds_f, s, P, d, rhs, x); \
    }
    // (few window sizes: every instantiation is a fully unrolled kernel and this file dominates the build time)
#ifdef DDH_SWEEP_ABLATE
    // timing ablations (build with -DDDH_SWEEP_ABLATE, select with DDH_ABL = bit mask; results are NOT a solve):
    //   1 forward: right-hand-side terms from lane-contiguous addresses   2 backward: lane-contiguous stores
    //   4 forward: one multiplier load per row                            8 backward: one factor load per row
    // Round 4, 512 x 512 x 256, solve ms per launch (gpurun_out/r4i): 0: 6.77 | 1: 5.36 | 2: 6.04 | 4: 6.77 | 8: 5.62 | all: 3.30
    static const int abl = getenv("DDH_ABL") ? atoi(getenv("DDH_ABL")) : 0;
    const_cast<RhsSrc &>(rhs).abl = abl;
    static bool abl_set = false;
    if (!abl_set) { (void)hipMemcpyToSymbol(HIP_SYMBOL(c_abl), &abl, sizeof(int)); abl_set = true; }
#endif
    bool lean_fwd = false;
    // one thread per (system, block) where the sweep kernels support it (LuDev::nsplit)
    const unsigned blocks_split = (unsigned)(((long)d.nsplit * d.Gp + 255) / 256);
    if constexpr (NF == 2) {
        lean_fwd = !use_fwd && lean_forward_ok(d);
        if (lean_fwd) {
#define DDH_LFWD(KLTV, NBTV) \
    { hipLaunchKernelGGL((solve_forward_lean_kernel<KLTV, NBTV>), dim3(blocks_split), dim3(256), lds_f, s, P, d, rhs, x); }
            // (block-parallel sweeps double the thread count: 4 waves per SIMD keep every thread resident, DDH_SWEEP_OCC=4)
            static const int occ4 = getenv("DDH_SWEEP_OCC") ? atoi(getenv("DDH_SWEEP_OCC")) == 4 : 1;
            if (forward_window(d.kl) == 6 && d.nb == 1 && rhs.n >= 1 && rhs.n <= 4 && sweep_deep(d))     // few systems: PD rows of loads in flight
                hipLaunchKernelGGL((solve_forward_deep_kernel<6, 1, 4, 4>), dim3(blocks_split * (256 / deep_block((long)d.nsplit * d.Gp))), dim3(deep_block((long)d.nsplit * d.Gp)), lds_f, s, P, d, rhs, x);
            else if (forward_window(d.kl) == 6 && d.nsplit > 1 && occ4 && d.nb <= 1)
                hipLaunchKernelGGL((solve_forward_lean_kernel<6, 1, 4>), dim3(blocks_split), dim3(256), lds_f, s, P, d, rhs, x);
            else if (forward_window(d.kl) == 6 && d.nb <= 1) DDH_LFWD(6, 1)
            else if (forward_window(d.kl) == 6) DDH_LFWD(6, 2) else DDH_LFWD(12, 2)
#undef DDH_LFWD
            if (d.nsplit > 1 && d.nb > 0)
                hipLaunchKernelGGL(border_finish_kernel<NF>, dim3(blocks), dim3(256), 0, s, P, d, x);
        }
    }
    if (use_fwd || lean_fwd) {
    } else if (d.nb <= 2) {
        if (d.kl <= 12) DDH_FWD(12, 2) else DDH_FWD(16, 2)
    } else {
        if (d.kl <= 12) DDH_FWD(12, 8) else DDH_FWD(16, 8)
    }
#undef DDH_FWD
    if (cb) {
#define DDH_CBWD(TTV, CBV)                                                                                         \
    {                                                                                                              \
        const unsigned cblocks = (d.real && d.nsplit > 1) ? (unsigned)(((long)d.nsplit * d.Gp) / (256 / CBV))      \
                                                          : (unsigned)((P.G + (256 / CBV) - 1) / (256 / CBV));     \
        if (d.real)                                                                                                \
            hipLaunchKernelGGL((solve_backward_coop_kernel<NF, TTV, true, CBV>), dim3(cblocks), dim3(256), lds_b, s, P, d, x); \
        else                                                                                                       \
            hipLaunchKernelGGL((solve_backward_coop_kernel<NF, TTV, false, CBV>), dim3(cblocks), dim3(256), lds_b, s, P, d, x); \
    }
        const int tt = (W + cb - 1) / cb;
        if constexpr (NF > 0) {
            if (cb == 16) {
                if (tt <= 2) DDH_CBWD(2, 16) else DDH_CBWD(3, 16)
            } else {
                if (tt <= 9) DDH_CBWD(9, 4) else DDH_CBWD(12, 4)
            }
        }
#undef DDH_CBWD
    }
#define DDH_SOLVE(WTV)                                                                                             \
    {                                                                                                              \
        if (d.real)                                                                                                \
            hipLaunchKernelGGL((solve_backward_kernel<NF, WTV, true, false>), dim3(blocks_split), dim3(256), lds_b, s, P, d, x, d.pband, rhs.skip); \
        else                                                                                                       \
            hipLaunchKernelGGL((solve_backward_kernel<NF, WTV, false, false>), dim3(blocks), dim3(256), lds_b, s, P, d, x, d.pband, rhs.skip); \
    }
    // (window sizes: few instantiations -- every one is a fully unrolled kernel and this file dominates the build time)
    bool fuse_p = false;
    if constexpr (NF == 2) fuse_p = want_p && d.real && d.pband != nullptr && d.n > 0 && !cb && W <= 48;
#define DDH_SOLVE_P(WTV)                                                                                           \
    { hipLaunchKernelGGL((solve_backward_kernel<NF, WTV, true, false, true>), dim3(blocks_split), dim3(256), lds_b, s, P, d, x, d.pband, rhs.skip); }
    if (fuse_p) {
        if constexpr (NF == 2) {
            static const int occ4 = getenv("DDH_SWEEP_OCC") ? atoi(getenv("DDH_SWEEP_OCC")) == 4 : 1;
            // DDH_BWD_RING = D: factor rows and y through a per-wave LDS-DMA ring D rows deep (solve_backward_ring_kernel);
            // 0 = direct loads.  LDS per workgroup: the permutation tables + 4 waves x D x 3.25 KiB.
            static const int ring = getenv("DDH_BWD_RING") ? atoi(getenv("DDH_BWD_RING")) : 0;
            const size_t tab = (((size_t)d.n * (d.pair ? 10 : 6)) + 1023) & ~(size_t)1023;
            if (W <= 17 && d.nsplit > 1 && occ4 && d.pair && ring >= 2 && ring <= 4 && P.G % 4 == 0) {
                const size_t lds_r = tab + (size_t)4 * ring * BwdRing<17>::SLOT;
#define DDH_RING(DV, MW)                                                                                                   \
    {                                                                                                                      \
        auto kern = solve_backward_ring_kernel<17, DV, MW>;                                                                \
        if (lds_r > 64 * 1024)                                                                                             \
            DDH_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));      \
        hipLaunchKernelGGL(kern, dim3(blocks_split), dim3(256), lds_r, s, P, d, x, d.pband, rhs.skip);                     \
    }
                if (ring == 2) DDH_RING(2, 4)
                else if (ring == 3) DDH_RING(3, 3)
                else DDH_RING(4, 2)
#undef DDH_RING
                // the waves whose lane quads do not share a factorization (unpaired cells, a ragged last wave)
                hipLaunchKernelGGL((solve_backward_kernel<NF, 17, true, false, true, 64, 4>), dim3(blocks_split), dim3(256), lds_b, s, P, d, x, d.pband, rhs.skip);
            } else if (W <= 17 && sweep_deep(d)) {
                // few systems (less than two waves per SIMD): PD rows of loads in flight in registers
                static const int pd = getenv("DDH_BWD_DEEP_PD") ? atoi(getenv("DDH_BWD_DEEP_PD")) : DDH_BWD_DEEP_PD;
                if (pd == 4)
                    hipLaunchKernelGGL((solve_backward_deep_kernel<17, 4>), dim3(blocks_split * (256 / deep_block((long)d.nsplit * d.Gp))), dim3(deep_block((long)d.nsplit * d.Gp)), lds_b, s, P, d, x, d.pband, rhs.skip);
                else
                    hipLaunchKernelGGL((solve_backward_deep_kernel<17, 2>), dim3(blocks_split * (256 / deep_block((long)d.nsplit * d.Gp))), dim3(deep_block((long)d.nsplit * d.Gp)), lds_b, s, P, d, x, d.pband, rhs.skip);
            } else if (W <= 17 && d.nsplit > 1 && occ4 && getenv("DDH_BWD_DBG") && atoi(getenv("DDH_BWD_DBG")) == 128)
                hipLaunchKernelGGL((solve_backward_kernel<NF, 17, true, false, true, 128, 4>), dim3(blocks_split), dim3(256), lds_b, s, P, d, x, d.pband, rhs.skip);
            else if (W <= 17 && d.nsplit > 1 && occ4)
                hipLaunchKernelGGL((solve_backward_kernel<NF, 17, true, false, true, 0, 4>), dim3(blocks_split), dim3(256), lds_b, s, P, d, x, d.pband, rhs.skip);
            else if (W <= 17) DDH_SOLVE_P(17)
            else if (W <= 32) DDH_SOLVE_P(32)
#ifdef DDH_BWD_ABLATE
            // timing ablations of the backward sweep (build with -DDDH_BWD_ABLATE; DDH_BWD_DBG = mask: 1 one factor load
            // per row, 2 no FMAs, 4 no stores, 8 no scratch load, 16 no window shift, 32 stores to lane-contiguous
            // addresses).  Results are NOT a solve.  Round 3: stores 2.2 ms, factor loads 2.9 ms of
            // the 4.9 ms sweep, FMAs and window shifts 0.
            else if (W <= 34 && getenv("DDH_BWD_DBG") && atoi(getenv("DDH_BWD_DBG")) > 0) {
#define DDH_SOLVE_DBG(V) case V: hipLaunchKernelGGL((solve_backward_kernel<NF, 34, true, false, true, V>), dim3(blocks_split), dim3(256), lds_b, s, P, d, x, d.pband, rhs.skip); break;
                switch (atoi(getenv("DDH_BWD_DBG"))) {
                    DDH_SOLVE_DBG(1) DDH_SOLVE_DBG(2) DDH_SOLVE_DBG(4) DDH_SOLVE_DBG(8) DDH_SOLVE_DBG(16) DDH_SOLVE_DBG(32) DDH_SOLVE_DBG(31)
                    default: DDH_SOLVE_P(34)
                }
#undef DDH_SOLVE_DBG
            }
#endif
            else if (W <= 34) DDH_SOLVE_P(34)
            else DDH_SOLVE_P(48)
        }
    } else if (d.n > 0 && !cb) {
        if (W <= 17) DDH_SOLVE(17)
        else if (W <= 32) DDH_SOLVE(32)
        else if (W <= 34) DDH_SOLVE(34)
        else if (W <= 48) DDH_SOLVE(48)
        else DDH_SOLVE(64)
    }
#undef DDH_SOLVE_P
#undef DDH_SOLVE
    DDH_HIP(hipGetLastError());
    if (did_p) *did_p = fuse_p;
    return finish_solve<NF>(pp, lu, rhs, x, s, fuse_p ? 1 : 0);
}


// Per-row upper width of the factors: wrow[j] = max over all factorizations of the last non-zero offset d of
// U(j, j + d).  Partial pivoting can fill up to kl extra super-diagonals, but rows where no factorization
// actually interchanged keep the original ku: the backward sweep then skips the all-zero tail entries.
constexpr int LUW_ROWS = 32;      // rows per workgroup of lu_width_kernel

template <bool REAL>
__global__ void __launch_bounds__(256)
lu_width_kernel(LuDev L, int *__restrict__ wrow, long only_blk = -1) {     // only_blk >= 0: that block of 64 factorizations alone (diagnostic)
    typedef typename El<REAL>::T E;
    const long gl = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = gl < L.GL && (only_blk < 0 || (gl >> 6) == only_blk);
    const E *Aw = (const E *)L.Aw;
    // blockIdx.y: a chunk of LUW_ROWS rows (a thread's scan of a row is a chain of dependent loads: the chunks run side by side)
    const int j0 = (int)blockIdx.y * LUW_ROWS, j1 = min(L.n, j0 + LUW_ROWS);
    for (int j = j0; j < j1; ++j) {
        int w = 0;
        if (act) {
            for (int d = L.W; d > 0; --d)
                if (!El<REAL>::is_zero(Aw[lu_aw(L, gl, j, L.kl + d)])) { w = d; break; }
        }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) w = max(w, __shfl_xor(w, sft, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(&wrow[j], w);
    }
}

}  // namespace ddh

using namespace ddh;

extern "C" {

int ddh_pencil_create(ddh_handle *pack, const ddh_pencil_geom *geom) {
    if (!geom || geom->nfourier < 0 || geom->nfourier > 2) return fail("pencil_create: nfourier must be 0, 1 or 2");
    PencilPack *pp = new PencilPack();
    pp->kind = H_PENCIL;
    PencilDev &d = pp->dev;
    d.nf = geom->nfourier;
    d.S = (d.nf == 2) ? 2 : 1;
    d.nrows = geom->nrows;
    d.nx = geom->nx;
    d.ny = geom->ny;
    if (d.nf == 0) {
        d.ncx = d.ncy = 1;
        if (d.nx != 1 || d.ny != 1) { delete pp; return fail("pencil_create: nfourier=0 needs nx=ny=1"); }
    } else if (d.nf == 1) {
        if (d.nx % 2 || d.ny != 1) { delete pp; return fail("pencil_create: nfourier=1 needs even nx and ny=1"); }
        d.ncx = d.nx / 2;
        d.ncy = 1;
    } else {
        if (d.nx % 2 || d.ny % 2) { delete pp; return fail("pencil_create: nfourier=2 needs even nx, ny"); }
        d.ncx = d.nx / 2;
        d.ncy = d.ny / 2;
    }
    d.ncells = d.ncx * d.ncy;
    d.G = d.ncells * d.S;
    d.mx_offset = geom->mx_offset;
    d.xtile = 0;
    int st = 0;
    if (d.nf >= 1) st = upload_vec(&pp->d_kx, geom->kx_h, (size_t)d.ncx);
    if (!st && d.nf == 2) st = upload_vec(&pp->d_ky, geom->ky_h, (size_t)d.ncy);
    if (st) { delete pp; return st; }
    d.kx = (const double *)pp->d_kx;
    d.ky = (const double *)pp->d_ky;
    if (d.nf >= 1) pp->kx_h.assign(geom->kx_h, geom->kx_h + d.ncx);
    if (d.nf == 2) pp->ky_h.assign(geom->ky_h, geom->ky_h + d.ncy);
    if (const char *e = getenv("DDH_SOLVE_COOP")) pp->coop_mode = atoi(e);
    if (const char *e = getenv("DDH_COOP_FWD")) pp->coop_fwd = atoi(e);
    if (const char *e = getenv("DDH_COOP_CB")) pp->coop_cb = atoi(e);
    *pack = register_handle(pp);
    return 0;
}

int ddh_pencil_set_solve_variant(ddh_handle pack, int mode, int fwd, int backward_lanes) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (mode < 0 || mode > 2) return fail("pencil_set_solve_variant: mode must be 0, 1 or 2");
    if (backward_lanes > 0 && backward_lanes != 4 && backward_lanes != 16)
        return fail("pencil_set_solve_variant: backward lanes per system must be 0, 4 or 16");
    pp->coop_mode = mode;
    pp->coop_fwd = fwd < 0 ? -1 : (fwd ? 1 : 0);
    pp->coop_cb = backward_lanes < 0 ? -1 : backward_lanes;
    return 0;
}

int ddh_pencil_set_pairing(ddh_handle pack, const int *row_swap_h, const int *col_swap_h, long min_systems) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    pp->pair_rows.clear();
    pp->pair_cols.clear();
    if (!row_swap_h || !col_swap_h) return 0;
    const int N = pp->dev.nrows;
    if (pp->dev.nf != 2) return fail("pencil_set_pairing: two Fourier axes required");
    for (int i = 0; i < N; ++i) {
        const int r = row_swap_h[i], c = col_swap_h[i];
        if (r < 0 || r >= N || c < 0 || c >= N || row_swap_h[r] != i || col_swap_h[c] != i)
            return fail("pencil_set_pairing: the row and column maps must be involutions of 0..nrows-1");
    }
    pp->pair_rows.assign(row_swap_h, row_swap_h + N);
    pp->pair_cols.assign(col_swap_h, col_swap_h + N);
    pp->pair_min = min_systems;
    return 0;
}

int ddh_pencil_set_row_blocks(ddh_handle pack, int nblocks) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (nblocks < 1 || nblocks > 8) return fail("pencil_set_row_blocks: 1 .. 8 blocks");
    pp->row_blocks = nblocks;
    return 0;
}

int ddh_pencil_add_matrix(ddh_handle pack, const ddh_polymat *mat, int nrows_out, int *mat_id) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    const int nt = mat->nterms;
    std::vector<int> order(nt);
    for (int i = 0; i < nt; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return mat->row_h[a] < mat->row_h[b]; });
    Matrix *m = new Matrix();
    std::vector<int> rowptr(nrows_out + 1, 0);
    m->col_h.resize(nt);
    m->coef_h.resize(nt);
    m->expo_h.resize(nt);
    m->row_h.resize(nt);
    for (int k = 0; k < nt; ++k) {
        const int t = order[k];
        const int r = mat->row_h[t];
        if (r < 0 || r >= nrows_out || mat->col_h[t] < 0 || mat->col_h[t] >= pp->dev.nrows) {
            delete m;
            return fail("pencil_add_matrix: term index out of range");
        }
        if (mat->ex_h[t] < 0 || mat->ex_h[t] > 7 || mat->ey_h[t] < 0 || mat->ey_h[t] > 7) {
            delete m;
            return fail("pencil_add_matrix: exponents must be in 0..7");
        }
        rowptr[r + 1]++;
        m->row_h[k] = r;
        m->col_h[k] = mat->col_h[t];
        m->coef_h[k] = make_double2(mat->coef_re_h[t], mat->coef_im_h[t]);
        m->expo_h[k] = (unsigned)mat->ex_h[t] | ((unsigned)mat->ey_h[t] << 8) |
                       ((unsigned)(mat->dx_h[t] ? 1 : 0) << 16) | ((unsigned)(mat->dy_h[t] ? 1 : 0) << 24);
    }
    for (int r = 0; r < nrows_out; ++r) rowptr[r + 1] += rowptr[r];
    int st = upload_vec(&m->d_rowptr, rowptr.data(), rowptr.size());
    if (!st) st = upload_vec(&m->d_col, m->col_h.data(), (size_t)nt);
    if (!st) st = upload_vec(&m->d_coef, m->coef_h.data(), (size_t)nt);
    if (!st) st = upload_vec(&m->d_expo, m->expo_h.data(), (size_t)nt);
    if (st) { delete m; return st; }
    m->dev.nrows_out = nrows_out;
    m->dev.nterms = nt;
    m->dev.rowptr = (const int *)m->d_rowptr;
    m->dev.col = (const int *)m->d_col;
    m->dev.coef = (const double2 *)m->d_coef;
    m->dev.expo = (const unsigned *)m->d_expo;
    m->dev.order = nullptr;
    // Row processing order for locality: with the rows of the system vector laid out [component][kz] a row (c, kz)
    // reads the columns (c', kz .. kz + band) of several components c'.  Processing the rows component by component
    // touches every column once per coupled component (x is re-read ~nnz/row times); processing them kz-major makes
    // all uses of a column fall into one short window of one thread block.  nz comes from DDH_MV_LOCALITY (experiment).
    {
        const char *e = getenv("DDH_MV_LOCALITY");
        const int nz = e ? atoi(e) : 0;
        if (nz > 1 && nrows_out >= 2 * nz) {
            const int nfield = (pp->dev.nrows / nz) * nz;
            std::vector<int> key(nrows_out, nz), ord(nrows_out);
            for (int k = 0; k < nt; ++k) {
                const int c = m->col_h[k];
                if (c < nfield) key[m->row_h[k]] = std::min(key[m->row_h[k]], c % nz);
            }
            for (int r = 0; r < nrows_out; ++r) ord[r] = r;
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return key[a] < key[b]; });
            if (!upload_vec(&m->d_order, ord.data(), ord.size())) m->dev.order = (const int *)m->d_order;
        }
    }
    // window form for band_matvec_kernel (two Fourier axes only): real, wavenumber-independent coefficients, the columns of
    // every row strictly ascending and inside MV_W consecutive rows
    m->dev.band = nullptr;
    m->dev.base = nullptr;
    if (pp->dev.nf == 2 && nt > 0 && !getenv("DDH_MV_NOBAND")) {
        bool ok = true;
        std::vector<double> band((size_t)nrows_out * MV_W, 0.0);
        std::vector<int> base(nrows_out, -1);
        for (int r = 0; r < nrows_out && ok; ++r) {
            for (int k = rowptr[r]; k < rowptr[r + 1] && ok; ++k) {
                if (m->expo_h[k] != 0 || m->coef_h[k].y != 0.0) ok = false;
                if (k > rowptr[r] && m->col_h[k] <= m->col_h[k - 1]) ok = false;
                if (k == rowptr[r]) base[r] = m->col_h[k];
                const int d = m->col_h[k] - base[r];
                if (d < 0 || d >= MV_W) ok = false;
                else band[(size_t)r * MV_W + d] = m->coef_h[k].x;
            }
        }
        if (ok && !upload_vec(&m->d_band, band.data(), band.size()) && !upload_vec(&m->d_base, base.data(), base.size())) {
            m->dev.band = (const double *)m->d_band;
            m->dev.base = (const int *)m->d_base;
        }
    }
    pp->mats.push_back(m);
    *mat_id = (int)pp->mats.size() - 1;
    return 0;
}

static int launch_matvec(PencilPack *pp, int mat_id, const double *x, double *y, const PostSolve &ps, void *stream,
                         int keep_empty = 0, int out_tiled = 0) {
    if (mat_id < 0 || mat_id >= (int)pp->mats.size()) return fail("pencil_matvec: bad matrix id");
    if (x == y) return fail("pencil_matvec: in-place unsupported");
    const PencilDev &P = pp->dev;
    const MatDev &A = pp->mats[mat_id]->dev;
    if (ps.nz > 0 && A.nrows_out % ps.nz) return fail("pencil_matvec_solve: rows are not a multiple of nz");
    const unsigned blocks = (unsigned)((P.ncells + 255) / 256);
    hipStream_t s = as_stream(stream);
    if (P.xtile && (P.nf != 2 || ps.nz > 0)) return fail("pencil_matvec: a tile-major state vector needs two Fourier axes and no fused back-substitution");
    if (ps.nz > 0 && blocks < 64 && !getenv("DDH_MV_FUSED_POST")) {
        // few cells: all rows in parallel first, then the light sequential recurrence (see postsolve_kernel)
        PostSolve none;
        memset(&none, 0, sizeof(none));
        const int st = launch_matvec(pp, mat_id, x, y, none, stream);
        if (st) return st;
        const dim3 g2((unsigned)((P.ncells + 63) / 64), (unsigned)(A.nrows_out / ps.nz));
        if (P.nf == 2)
            hipLaunchKernelGGL(postsolve_kernel<2>, g2, dim3(64), 0, s, P, y, ps);
        else if (P.nf == 1)
            hipLaunchKernelGGL(postsolve_kernel<1>, g2, dim3(64), 0, s, P, y, ps);
        else
            hipLaunchKernelGGL(postsolve_kernel<0>, g2, dim3(64), 0, s, P, y, ps);
        DDH_HIP(hipGetLastError());
        return 0;
    }
    int rpc;
    if (ps.nz > 0) {
        rpc = ps.nz;
    } else {
        static const int nch = getenv("DDH_MV_CHUNKS") ? atoi(getenv("DDH_MV_CHUNKS")) : 32;
        // few cells: more, shorter row chunks (the rows are independent) so that the launch still fills the chip
        const int want = (blocks >= 64) ? nch : (int)std::min<long>(512, 2048 / blocks);
        rpc = (A.nrows_out + want - 1) / want;
        if (rpc < 8) rpc = 8;
    }
    const unsigned chunks = (unsigned)((A.nrows_out + rpc - 1) / rpc);
    if (chunks > 65535) return fail("pencil_matvec: too many row chunks");
    const dim3 grid(blocks, chunks ? chunks : 1);
    if (P.nf == 2 && A.band && ps.nz == 0 && !A.order && (blocks >= 64 || out_tiled))
        hipLaunchKernelGGL(band_matvec_kernel, grid, dim3(256), 0, s, P, A, x, y, rpc, keep_empty, out_tiled);
    else if (out_tiled)
        return fail("pencil_matvec_update_tiled: only the window-form mat-vec (real, wavenumber-independent bands; two Fourier "
                    "axes) writes the tile-major layout");
    else if (P.nf == 2)
        hipLaunchKernelGGL(matvec_kernel<2>, grid, dim3(256), 0, s, P, A, x, y, ps, rpc);
    else if (P.nf == 1)
        hipLaunchKernelGGL(matvec_kernel<1>, grid, dim3(256), 0, s, P, A, x, y, ps, rpc);
    else
        hipLaunchKernelGGL(matvec_kernel<0>, grid, dim3(256), 0, s, P, A, x, y, ps, rpc);
    DDH_HIP(hipGetLastError());
    return 0;
}

/* The state vector X -- the solution vector of every solve and the input vector of every mat-vec of this pack from now on --
 * is stored tile-major ([kx / 8][ky / 8][kx % 8][ky % 8] within a row, like the right-hand-side vectors of
 * ddh_pencil_solve_recombined_tiled): a wavefront's stores of a solution row are two 512-byte runs instead of sixteen 64-byte
 * runs.  Two Fourier axes with storage sizes that are multiples of 8; on = 0 returns to the natural layout. */
int ddh_pencil_set_state_tiled(ddh_handle pack, int on) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (on && (pp->dev.nf != 2 || (pp->dev.nx & 7) || (pp->dev.ny & 7)))
        return fail("pencil_set_state_tiled: two Fourier axes with storage sizes that are multiples of 8");
    if (on < 0 || on > 2) return fail("pencil_set_state_tiled: 0 natural, 1 tile-major rows, 2 kx-band-major");
    pp->dev.xtile = on;
    return 0;
}

int ddh_pencil_matvec(ddh_handle pack, int mat_id, const double *x, double *y, void *stream) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    PostSolve ps;
    memset(&ps, 0, sizeof(ps));
    return launch_matvec(pp, mat_id, x, y, ps, stream);
}

int ddh_pencil_matvec_update(ddh_handle pack, int mat_id, const double *x, double *y, void *stream) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    PostSolve ps;
    memset(&ps, 0, sizeof(ps));
    return launch_matvec(pp, mat_id, x, y, ps, stream, 1);
}

int ddh_pencil_matvec_update_tiled(ddh_handle pack, int mat_id, const double *x, double *y, void *stream) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (pp->dev.nf != 2 || (pp->dev.nx & 7) || (pp->dev.ny & 7))
        return fail("pencil_matvec_update_tiled: two Fourier axes with storage sizes that are multiples of 8");
    PostSolve ps;
    memset(&ps, 0, sizeof(ps));
    return launch_matvec(pp, mat_id, x, y, ps, stream, 1, 1);
}

int ddh_pencil_add_upper_bands(ddh_handle pack, int nz, int nbands, const int *offsets_h, const double *bands_h,
                               int *bands_id) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (nz < 1 || nbands < 1 || nbands > 4 || offsets_h[0] != 0) return fail("pencil_add_upper_bands: bad band description");
    for (int d = 1; d < nbands; ++d)
        if (offsets_h[d] < 1 || offsets_h[d] > 4) return fail("pencil_add_upper_bands: band offsets must be 1..4");
    PostSolve ps;
    memset(&ps, 0, sizeof(ps));
    ps.nz = nz;
    ps.nbands = nbands;
    for (int d = 0; d < nbands; ++d) ps.off[d] = offsets_h[d];
    void *mem = nullptr;
    int st = upload_vec(&mem, bands_h, (size_t)nbands * nz);
    if (st) return st;
    ps.bands = (const double *)mem;
    pp->post_mem.push_back(mem);
    pp->posts.push_back(ps);
    *bands_id = (int)pp->posts.size() - 1;
    return 0;
}

int ddh_pencil_matvec_solve(ddh_handle pack, int mat_id, int bands_id, const double *x, double *y, void *stream) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (bands_id < 0 || bands_id >= (int)pp->posts.size()) return fail("pencil_matvec_solve: bad bands id");
    return launch_matvec(pp, mat_id, x, y, pp->posts[bands_id], stream);
}

static int factor_impl(ddh_handle pack, int matM_id, int matL_id, double a, double b, const int *row_perm_h,
                       const int *col_perm_h, int n_interior, int kl, int ku, const unsigned char *row_axes_h,
                       const unsigned char *col_axes_h, const unsigned char *row_code_h,
                       const unsigned char *col_code_h, int reuse_lu_id, int *lu_id, void *stream) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    const PencilDev &P = pp->dev;
    const bool real = (row_code_h != nullptr && col_code_h != nullptr);
    const int nm = (int)pp->mats.size();
    if (matM_id < 0 || matM_id >= nm || matL_id < 0 || matL_id >= nm) return fail("pencil_factor: bad matrix id");
    const int N = P.nrows, n = n_interior, nb = N - n;
    if (n < 0 || nb < 0) return fail("pencil_factor: bad interior size");
    if (kl > KLMAX) return fail("pencil_factor: lower bandwidth " + std::to_string(kl) + " exceeds KLMAX=16");
    if (nb > NBMAX) return fail("pencil_factor: border size " + std::to_string(nb) + " exceeds NBMAX=8");
    const int W = ku + kl;
    if (W > 64) return fail("pencil_factor: band too wide (ku+kl=" + std::to_string(W) + " > 64)");
    if (nb > W && n > 0) return fail("pencil_factor: border wider than the band window");
    if (pp->mats[matM_id]->dev.nrows_out != N || pp->mats[matL_id]->dev.nrows_out != N)
        return fail("pencil_factor: matrices must be square (nrows x nrows)");
    LuFactor *lu = nullptr;
    const bool reuse = reuse_lu_id >= 0 && reuse_lu_id < (int)pp->lus.size() && pp->lus[reuse_lu_id] &&
                       pp->lus[reuse_lu_id]->dev.n == n && pp->lus[reuse_lu_id]->dev.kl == kl &&
                       pp->lus[reuse_lu_id]->dev.ku == ku && (pp->lus[reuse_lu_id]->dev.real != 0) == real;
    hipStream_t s = as_stream(stream);
    // partner pencils: cells (mx, my) and (my, mx), mx != my, both off the axes, share one stored factorization
    std::vector<long> vcell, slot_cells;
    std::vector<int> vslot;
    bool pair = real && P.nf == 2 && !pp->pair_rows.empty() && P.G >= pp->pair_min && P.ncx == P.ncy &&
                P.mx_offset == 0 && pp->kx_h == pp->ky_h && n > 0;
    if (reuse && (pp->lus[reuse_lu_id]->dev.pair != 0) != pair) return fail("pencil_factor: pairing changed under a reused LU");
    if (pair) {
        // pairs first, in 4 x 4 tiles of cells so that the 16 pairs of a wavefront touch 64-byte runs of the system vectors
        // in both the cell's and the partner's rows; the four wavefronts of a workgroup take the 2 x 2 tiles of an 8 x 8
        // super-tile where all four are full (DDH_PAIR_TILE8, default on), so that the workgroup's accesses cover whole
        // 128-byte lines in both rows; then the remaining tiles, then the unpaired cells (axes, diagonal) in cell order
        const long nc = P.ncx, nt = (nc + 3) / 4;
        static const int tile8 = getenv("DDH_PAIR_TILE8") ? atoi(getenv("DDH_PAIR_TILE8")) : 1;
        auto has_partner = [&](long mx, long my) { return mx > 0 && my > 0 && mx != my; };
        std::vector<char> done((size_t)(nt * nt), 0);
        auto emit_tile = [&](long tx, long ty) {
            done[(size_t)(tx * nt + ty)] = 1;
            for (long mx = 4 * tx; mx < std::min(nc, 4 * tx + 4); ++mx)
                for (long my = 4 * ty; my < std::min(nc, 4 * ty + 4); ++my) {
                    if (!has_partner(mx, my) || mx > my) continue;
                    const int slot = (int)(slot_cells.size() / 2);
                    slot_cells.push_back(mx * P.ncy + my);
                    slot_cells.push_back(my * P.ncy + mx);
                    vcell.push_back(mx * P.ncy + my);
                    vslot.push_back(2 * slot);
                    vcell.push_back(my * P.ncy + mx);
                    vslot.push_back(2 * slot + 1);
                }
        };
        if (tile8 && nc % 8 == 0) {
            const long ns = nt / 2;
            for (long sx = 1; sx < ns; ++sx)               // (sx = 0 holds the mx = 0 axis: its tiles are not full)
                for (long sy = sx + 1; sy < ns; ++sy)
                    for (long tx = 2 * sx; tx < 2 * sx + 2; ++tx)
                        for (long ty = 2 * sy; ty < 2 * sy + 2; ++ty) emit_tile(tx, ty);
        }
        for (long tx = 0; tx < nt; ++tx)
            for (long ty = tx; ty < nt; ++ty)
                if (!done[(size_t)(tx * nt + ty)]) emit_tile(tx, ty);
        for (long mx = 0; mx < nc; ++mx)
            for (long my = 0; my < nc; ++my) {
                if (has_partner(mx, my)) continue;
                const int slot = (int)(slot_cells.size() / 2);
                slot_cells.push_back(mx * P.ncy + my);
                slot_cells.push_back(-1);
                vcell.push_back(mx * P.ncy + my);
                vslot.push_back(2 * slot);
            }
    }
    const size_t G = (size_t)P.G;   // threads of a solve / scratch stride
    const size_t GL = pair ? slot_cells.size() / 2 : (real ? (size_t)P.ncells : (size_t)P.G);
    const size_t esz = real ? sizeof(double) : sizeof(double2);
    if (reuse) {
        lu = pp->lus[reuse_lu_id];
        lu->d_binv = nullptr;       // an explicit block inverse belongs to the OLD (a, b): the caller registers a new one
    } else {
        lu = new LuFactor();
        LuDev &d = lu->dev;
        memset(&d, 0, sizeof(d));
        d.n = n; d.nb = nb; d.N = N; d.kl = kl; d.ku = ku; d.W = W; d.BW = kl + W + 1;
        // real factors: rows are padded (with zeros) to the register window of the one-thread-per-system backward kernel,
        // which then runs without per-entry guards (launch_solve picks the same window)
        if (real) {
            d.kpad = forward_window(kl) - kl;                       // the diagonal lands on an even entry (12 or 16)
            d.BW = d.kpad + kl + backward_window(W) + 1;
            d.BW += d.BW & 1;
            d.pk = 1;       // (neutral for the cooperative sweeps: 2-D RB 512 x 256 steps at 560 steps/s either way)
            d.pk63 = 63;
        }
        d.real = real ? 1 : 0;
        d.nsplit = 1;
        d.nh = n;
        d.Gp = (long)((G + 255) / 256) * 256;
        static const bool no_par = getenv("DDH_SPLIT_THREADS") && atoi(getenv("DDH_SPLIT_THREADS")) == 0;
        if (real && P.nf >= 1 && pp->row_blocks > 1 && n > 0 && n % pp->row_blocks == 0 && !no_par) {
            d.nsplit = pp->row_blocks;
            d.nh = n / pp->row_blocks;
        }
        d.GL = (long)GL;
        d.nblk = (long)((GL + 63) / 64);
        d.rows_aw = (n > 0 ? n : 1) + (real ? forward_window(kl) : 0);
        const size_t GLp = (size_t)d.nblk * 64;
        const size_t szAw = esz * (size_t)d.rows_aw * d.BW * GLp;
        const size_t szAb = esz * (size_t)N * (nb > 0 ? nb : 1) * GLp;
        const size_t szScr = sizeof(double2) * (size_t)std::max(n + nb + d.nsplit * nb, nb * nb) * G;
        int st = check_hip(hipMalloc((void **)&d.Aw, szAw), "hipMalloc(band LU)");
        if (!st) st = check_hip(hipMalloc((void **)&d.Ab, szAb), "hipMalloc(border LU)");
        if (!st) st = check_hip(hipMalloc((void **)&d.piv, (size_t)d.rows_aw * GLp), "hipMalloc(piv)");
        if (!st) st = check_hip(hipMalloc((void **)&d.flag, GL), "hipMalloc(flag)");
        if (!st) st = check_hip(hipMalloc((void **)&d.scratch, szScr), "hipMalloc(scratch)");
        if (!st) st = upload_vec(&lu->d_rowperm, row_perm_h, (size_t)N);
        if (!st) st = upload_vec(&lu->d_colperm, col_perm_h, (size_t)N);
        lu->colperm_h.assign(col_perm_h, col_perm_h + N);
        if (real) lu->ccode_h.assign(col_code_h, col_code_h + N);
        d.pband = nullptr;
        if (!st) st = upload_vec(&lu->d_raxes, row_axes_h + n, (size_t)nb);
        if (!st) st = upload_vec(&lu->d_caxes, col_axes_h + n, (size_t)nb);
        if (!st && real) st = upload_vec(&lu->d_rcode, row_code_h, (size_t)N);
        if (!st && real) st = upload_vec(&lu->d_ccode, col_code_h, (size_t)N);
        if (!st && pair) {
            std::vector<int> rp2(N), cp2(N);
            for (int i = 0; i < N; ++i) {
                rp2[i] = pp->pair_rows[row_perm_h[i]];
                cp2[i] = pp->pair_cols[col_perm_h[i]];
            }
            std::vector<long> sc(GL);
            for (size_t i = 0; i < GL; ++i) sc[i] = slot_cells[2 * i];
            st = upload_vec(&lu->d_vcell, vcell.data(), vcell.size());
            if (!st) st = upload_vec(&lu->d_vslot, vslot.data(), vslot.size());
            if (!st) st = upload_vec(&lu->d_slot_cell, sc.data(), sc.size());
            if (!st) st = upload_vec(&lu->d_rowperm2, rp2.data(), (size_t)N);
            if (!st) st = upload_vec(&lu->d_colperm2, cp2.data(), (size_t)N);
            lu->slot_cells_h = slot_cells;
            d.pair = 1;
            d.vcell = (const long *)lu->d_vcell;
            d.vslot = (const int *)lu->d_vslot;
            d.slot_cell = (const long *)lu->d_slot_cell;
            d.rowperm2 = (const int *)lu->d_rowperm2;
            d.colperm2 = (const int *)lu->d_colperm2;
        }
        if (st) { free_lu(lu); return st; }
        d.rowperm = (const int *)lu->d_rowperm;
        d.colperm = (const int *)lu->d_colperm;
        d.row_axes = (const unsigned char *)lu->d_raxes;
        d.col_axes = (const unsigned char *)lu->d_caxes;
        d.row_code = (const unsigned char *)lu->d_rcode;
        d.col_code = (const unsigned char *)lu->d_ccode;
        lu->bytes = szAw + szAb + szScr + (size_t)n * GL + GL;
        // interior rows / columns must exist for every cell
        for (int i = 0; i < n; ++i)
            if ((row_axes_h[i] & 3) != 3 || (col_axes_h[i] & 3) != 3) {
                free_lu(lu);
                return fail("pencil_factor: interior rows/columns must be valid for all pencils");
            }
    }
    LuDev &d = lu->dev;
    DDH_HIP(hipMemsetAsync(d.Aw, 0, esz * (size_t)d.rows_aw * d.BW * (size_t)d.nblk * 64, s));
    DDH_HIP(hipMemsetAsync(d.Ab, 0, esz * (size_t)N * (nb > 0 ? nb : 1) * (size_t)d.nblk * 64, s));
    // inverse permutations (physical -> logical) on the device
    std::vector<int> rowinv(N), colinv(N);
    for (int i = 0; i < N; ++i) {
        rowinv[row_perm_h[i]] = i;
        colinv[col_perm_h[i]] = i;
    }
    void *d_rowinv = nullptr, *d_colinv = nullptr;
    int st = upload_vec(&d_rowinv, rowinv.data(), (size_t)N);
    if (!st) st = upload_vec(&d_colinv, colinv.data(), (size_t)N);
    if (st) { if (!reuse) free_lu(lu); return st; }
    const unsigned blocks = (unsigned)((GL + 63) / 64);
    // rows of an elimination step spread over the workgroup (factor_rows_kernel) when kl + 1 + nb row threads fit one;
    // DDH_FACTOR_ROWS=0: one thread per system (factor_kernel)
    static const bool rows_off = getenv("DDH_FACTOR_ROWS") && atoi(getenv("DDH_FACTOR_ROWS")) == 0;
    const int NT = d.kl + 1 + d.nb;
    static const int rows_mode = getenv("DDH_FACTOR_ROWS") ? atoi(getenv("DDH_FACTOR_ROWS")) : 2;   // 2: register window
    if (!rows_off && rows_mode >= 2 && real && NT <= FR_NTMAX && GL >= 64 && d.W + 1 <= FW_WMAX) {
        // (real factors: 72 of the 128 registers a 14-wave workgroup allows; the complex window would spill)
        static const bool narrow_ok = !(getenv("DDH_FACTOR_NARROW") && atoi(getenv("DDH_FACTOR_NARROW")) == 0);
        if (narrow_ok && d.W + 1 <= FW_WNARROW && NT <= FW_NT_NARROW) {
            // three workgroups per CU instead of two: the 514 workgroups of 512^2 pencils run in ONE round (see FW_WNARROW)
            const size_t lds = 2 * (size_t)FW_WNARROW * 64 * esz;
            hipLaunchKernelGGL((factor_window_kernel<true, FW_WNARROW, FW_NT_NARROW, 6>), dim3(blocks), dim3(64, NT), lds, s, P, d,
                               pp->mats[matM_id]->dev, pp->mats[matL_id]->dev, a, b, (const int *)d_rowinv, (const int *)d_colinv);
        } else {
            const size_t lds = 2 * (size_t)FW_WMAX * 64 * esz;
            hipLaunchKernelGGL(factor_window_kernel<true>, dim3(blocks), dim3(64, NT), lds, s, P, d, pp->mats[matM_id]->dev,
                               pp->mats[matL_id]->dev, a, b, (const int *)d_rowinv, (const int *)d_colinv);
        }
    } else if (!rows_off && NT <= FR_NTMAX && GL >= 64) {
        if (real)
            hipLaunchKernelGGL(factor_rows_kernel<true>, dim3(blocks), dim3(64, NT), 0, s, P, d, pp->mats[matM_id]->dev,
                               pp->mats[matL_id]->dev, a, b, (const int *)d_rowinv, (const int *)d_colinv);
        else
            hipLaunchKernelGGL(factor_rows_kernel<false>, dim3(blocks), dim3(64, NT), 0, s, P, d, pp->mats[matM_id]->dev,
                               pp->mats[matL_id]->dev, a, b, (const int *)d_rowinv, (const int *)d_colinv);
    } else if (real)
        hipLaunchKernelGGL(factor_kernel<true>, dim3(blocks), dim3(64), 0, s, P, d, pp->mats[matM_id]->dev,
                           pp->mats[matL_id]->dev, a, b, (const int *)d_rowinv, (const int *)d_colinv);
    else
        hipLaunchKernelGGL(factor_kernel<false>, dim3(blocks), dim3(64), 0, s, P, d, pp->mats[matM_id]->dev,
                           pp->mats[matL_id]->dev, a, b, (const int *)d_rowinv, (const int *)d_colinv);
    DDH_HIP(hipGetLastError());
    DDH_HIP(hipStreamSynchronize(s));
    (void)hipFree(d_rowinv);
    (void)hipFree(d_colinv);
    // flagged factorizations -> flagged cells
    std::vector<unsigned char> flags(GL);
    DDH_HIP(hipMemcpy(flags.data(), d.flag, GL, hipMemcpyDeviceToHost));
    lu->flag_cells.clear();
    const int per = real ? 1 : P.S;
    if (d.pair) {
        for (size_t slot = 0; slot < GL; ++slot)
            if (flags[slot]) {
                lu->flag_cells.push_back(lu->slot_cells_h[2 * slot]);
                if (lu->slot_cells_h[2 * slot + 1] >= 0) lu->flag_cells.push_back(lu->slot_cells_h[2 * slot + 1]);
            }
        std::sort(lu->flag_cells.begin(), lu->flag_cells.end());
    } else {
        for (long cidx = 0; cidx < P.ncells; ++cidx) {
            bool f = false;
            for (int sidx = 0; sidx < per; ++sidx) f = f || flags[cidx * per + sidx];
            if (f) lu->flag_cells.push_back(cidx);
        }
    }
    lu->nflag = (int)lu->flag_cells.size();
    (void)hipFree(lu->d_flag_cells); lu->d_flag_cells = nullptr;
    (void)hipFree(lu->d_inv); lu->d_inv = nullptr;
    (void)hipFree(lu->d_dense_rhs); lu->d_dense_rhs = nullptr;
    if (lu->nflag) {
        st = upload_vec(&lu->d_flag_cells, lu->flag_cells.data(), lu->flag_cells.size());
        if (st) return st;
    }
    // the fill of U, row by row, for the backward sweep (LuDev::wrow); DDH_BWD_ROW_FILL=0: every pair is loaded
    static const bool row_fill = !(getenv("DDH_BWD_ROW_FILL") && atoi(getenv("DDH_BWD_ROW_FILL")) == 0);
    lu->dev.wrow = nullptr;
    if (row_fill && real && n > 0) {
        if (!lu->d_wrow) DDH_HIP(hipMalloc(&lu->d_wrow, (size_t)n * sizeof(int)));
        DDH_HIP(hipMemsetAsync(lu->d_wrow, 0, (size_t)n * sizeof(int), s));
        hipLaunchKernelGGL(lu_width_kernel<true>, dim3((unsigned)((lu->dev.GL + 255) / 256), (unsigned)((n + LUW_ROWS - 1) / LUW_ROWS)),
                           dim3(256), 0, s, lu->dev, (int *)lu->d_wrow, -1L);
        DDH_HIP(hipGetLastError());
        lu->dev.wrow = (const int *)lu->d_wrow;
    }
    if (reuse) {
        *lu_id = reuse_lu_id;
    } else {
        pp->lus.push_back(lu);
        *lu_id = (int)pp->lus.size() - 1;
    }
    return 0;
}

int ddh_pencil_factor(ddh_handle pack, int matM_id, int matL_id, double a, double b, const int *row_perm_h,
                      const int *col_perm_h, int n_interior, int kl, int ku, const unsigned char *row_axes_h,
                      const unsigned char *col_axes_h, int reuse_lu_id, int *lu_id, void *stream) {
    return factor_impl(pack, matM_id, matL_id, a, b, row_perm_h, col_perm_h, n_interior, kl, ku, row_axes_h,
                       col_axes_h, nullptr, nullptr, reuse_lu_id, lu_id, stream);
}

int ddh_pencil_factor_real(ddh_handle pack, int matM_id, int matL_id, double a, double b, const int *row_perm_h,
                           const int *col_perm_h, int n_interior, int kl, int ku,
                           const unsigned char *row_axes_h, const unsigned char *col_axes_h,
                           const unsigned char *row_code_h, const unsigned char *col_code_h, int reuse_lu_id,
                           int *lu_id, void *stream) {
    if (!row_code_h || !col_code_h) return fail("pencil_factor_real: grading codes required");
    return factor_impl(pack, matM_id, matL_id, a, b, row_perm_h, col_perm_h, n_interior, kl, ku, row_axes_h,
                       col_axes_h, row_code_h, col_code_h, reuse_lu_id, lu_id, stream);
}

/* number of flagged cells of a factorization and their ids (host array of *count longs) */
int ddh_pencil_flagged(ddh_handle pack, int lu_id, int *count, long *cells_h, int max_cells) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (lu_id < 0 || lu_id >= (int)pp->lus.size()) return fail("pencil_flagged: bad LU id");
    LuFactor *lu = pp->lus[lu_id];
    *count = lu->nflag;
    for (int i = 0; i < lu->nflag && i < max_cells; ++i) cells_h[i] = lu->flag_cells[i];
    return 0;
}

/* explicit inverses (logical ordering, row-major N x N complex) for every system of every flagged
 * cell, in the order of ddh_pencil_flagged: inv_h[(f*S + s)*N*N + i*N + j] as interleaved re/im */
int ddh_pencil_set_dense_inverse(ddh_handle pack, int lu_id, const double *inv_h) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (lu_id < 0 || lu_id >= (int)pp->lus.size()) return fail("pencil_set_dense_inverse: bad LU id");
    LuFactor *lu = pp->lus[lu_id];
    if (!lu->nflag) return 0;
    const size_t N = (size_t)lu->dev.N, nsys = (size_t)lu->nflag * pp->dev.S;
    (void)hipFree(lu->d_inv);
    (void)hipFree(lu->d_dense_rhs);
    DDH_HIP(hipMalloc(&lu->d_inv, nsys * N * N * sizeof(double2)));
    DDH_HIP(hipMalloc(&lu->d_dense_rhs, 2 * nsys * N * sizeof(double2)));
    DDH_HIP(hipMemcpy(lu->d_inv, inv_h, nsys * N * N * sizeof(double2), hipMemcpyHostToDevice));
    return 0;
}

/* the same from DEVICE memory, one system at a time: inv_d = N x N row-major inverse of system `sys` (= f * S + s in the
 * order of ddh_pencil_flagged), real (is_complex = 0) or interleaved complex; widened / copied into the pack's storage
 * on the stream.  With ddh_dense_inverse_* a change of the timestep needs no host linear algebra for flagged pencils. */
__global__ void __launch_bounds__(256) widen_inverse_kernel(const double *__restrict__ src, double2 *__restrict__ dst, long nn,
                                                            int cx) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nn; e += (long)gridDim.x * 256)
        dst[e] = cx ? make_double2(src[2 * e], src[2 * e + 1]) : make_double2(src[e], 0.0);
}

int ddh_pencil_set_dense_inverse_dev(ddh_handle pack, int lu_id, int sys, const double *inv_d, int is_complex, void *stream) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (lu_id < 0 || lu_id >= (int)pp->lus.size()) return fail("pencil_set_dense_inverse_dev: bad LU id");
    LuFactor *lu = pp->lus[lu_id];
    const size_t N = (size_t)lu->dev.N, nsys = (size_t)lu->nflag * pp->dev.S;
    if (sys < 0 || (size_t)sys >= nsys) return fail("pencil_set_dense_inverse_dev: system index out of range");
    if (!lu->d_inv || !lu->d_dense_rhs) {
        // both or neither: a solve that finds d_inv expects its right-hand-side buffer too
        (void)hipFree(lu->d_inv); lu->d_inv = nullptr;
        (void)hipFree(lu->d_dense_rhs); lu->d_dense_rhs = nullptr;
        void *inv = nullptr, *rhs = nullptr;
        int st = check_hip(hipMalloc(&inv, nsys * N * N * sizeof(double2)), "hipMalloc(dense inverse)");
        if (!st) st = check_hip(hipMalloc(&rhs, 2 * nsys * N * sizeof(double2)), "hipMalloc(dense rhs)");
        if (st) {
            (void)hipFree(inv);
            return st;
        }
        lu->d_inv = inv;
        lu->d_dense_rhs = rhs;
    }
    hipLaunchKernelGGL(widen_inverse_kernel, dim3(2048), dim3(256), 0, as_stream(stream), inv_d,
                       (double2 *)lu->d_inv + (size_t)sys * N * N, (long)(N * N), is_complex);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_pencil_set_block_inverse(ddh_handle pack, int lu_id, const double *binv_d) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (lu_id < 0 || lu_id >= (int)pp->lus.size()) return fail("pencil_set_block_inverse: bad LU id");
    LuFactor *lu = pp->lus[lu_id];
    if (binv_d) {
        const LuDev &d = lu->dev;
        if (pp->dev.nf != 1 || !d.real || d.pair || d.nsplit < 1 || d.n != d.nsplit * d.nh || d.nh > 4096)     // (nh * 16 B of LDS per workgroup)
            return fail("pencil_set_block_inverse: one Fourier axis, real-graded factors, equal diagonal blocks");
    }
    lu->d_binv = binv_d;
    return 0;
}

int ddh_pencil_solve(ddh_handle pack, int lu_id, const double *rhs, double *x, void *stream) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (lu_id < 0 || lu_id >= (int)pp->lus.size()) return fail("pencil_solve: bad LU id");
    if (rhs == x) return fail("pencil_solve: in-place unsupported");
    const double one = 1.0;
    return ddh_pencil_solve_lincomb(pack, lu_id, 1, &rhs, &one, x, stream);
}

int ddh_pencil_solve_lincomb(ddh_handle pack, int lu_id, int nterms, const double *const *xs_h, const double *alpha_h,
                             double *x, void *stream) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (lu_id < 0 || lu_id >= (int)pp->lus.size()) return fail("pencil_solve: bad LU id");
    if (nterms < 1 || nterms > RHS_MAX) return fail("pencil_solve_lincomb: 1 to 8 right-hand-side terms");
    RhsSrc r;
    memset(&r, 0, sizeof(r));
    r.n = nterms;
    for (int t = 0; t < nterms; ++t) {
        if (xs_h[t] == x) return fail("pencil_solve: in-place unsupported");
        r.p[t] = xs_h[t];
        r.a[t] = alpha_h[t];
    }
    LuFactor *lu = pp->lus[lu_id];
    hipStream_t s = as_stream(stream);
    if (pp->dev.nf == 2) return launch_solve<2>(pp, lu, r, x, s);
    if (pp->dev.nf == 1) return launch_solve<1>(pp, lu, r, x, s);
    return launch_solve<0>(pp, lu, r, x, s);
}

// band table of the recombination matrix P (a registered constant real matrix) in the logical ordering of an LU;
// returns 0 and leaves lu->dev.pband null when P cannot be fused (not unit upper banded within PBW, complex or
// wavenumber-dependent entries, differing grading codes, border columns)
static int build_pband(PencilPack *pp, LuFactor *lu, int p_mat_id) {
    LuDev &d = lu->dev;
    if (lu->pband_mat == p_mat_id && lu->d_pband) { d.pband = (const double *)lu->d_pband; return 0; }
    d.pband = nullptr;
    lu->pband_mat = p_mat_id;
    if (!d.real || d.n <= 0 || lu->colperm_h.empty()) return 0;
    const Matrix *m = pp->mats[p_mat_id];
    const int N = d.N, n = d.n;
    if (m->dev.nrows_out != N) return 0;
    std::vector<int> cinv(N, -1);
    for (int i = 0; i < N; ++i) cinv[lu->colperm_h[i]] = i;
    std::vector<double> band((size_t)n * PBW, 0.0);
    std::vector<char> diag(N, 0);
    for (size_t t = 0; t < m->row_h.size(); ++t) {
        const int r = cinv[m->row_h[t]], c = cinv[m->col_h[t]];
        const double2 cf = m->coef_h[t];
        if (m->expo_h[t] != 0 || cf.y != 0.0) return 0;
        if (r == c) {
            if (cf.x != 1.0 || diag[r]) return 0;
            diag[r] = 1;
            continue;
        }
        if (r >= n || c >= n || c < r || c - r > PBW) return 0;
        if (lu->ccode_h[r] != lu->ccode_h[c]) return 0;
        band[(size_t)r * PBW + (c - r - 1)] += cf.x;
    }
    for (int i = 0; i < N; ++i)
        if (!diag[i]) return 0;
    if (!lu->d_pband) DDH_HIP(hipMalloc(&lu->d_pband, band.size() * sizeof(double) + 16));
    DDH_HIP(hipMemcpy(lu->d_pband, band.data(), band.size() * sizeof(double), hipMemcpyHostToDevice));
    d.pband = (const double *)lu->d_pband;
    return 0;
}

int ddh_pencil_solve_recombined(ddh_handle pack, int lu_id, int nterms, const double *const *xs_h,
                                const double *alpha_h, int p_mat_id, double *work, double *x, void *stream) {
    return ddh_pencil_solve_recombined_sparse(pack, lu_id, nterms, xs_h, alpha_h, p_mat_id, work, x, nullptr, nullptr, stream);
}

static int solve_recombined_impl(ddh_handle pack, int lu_id, int nterms, const double *const *xs_h, const double *alpha_h,
                                 int p_mat_id, double *work, double *x, const unsigned char *zero_rows,
                                 const unsigned char *skip_rows, int terms_tiled, void *stream);

int ddh_pencil_solve_recombined_sparse(ddh_handle pack, int lu_id, int nterms, const double *const *xs_h,
                                       const double *alpha_h, int p_mat_id, double *work, double *x,
                                       const unsigned char *zero_rows, const unsigned char *skip_rows, void *stream) {
    return solve_recombined_impl(pack, lu_id, nterms, xs_h, alpha_h, p_mat_id, work, x, zero_rows, skip_rows, 0, stream);
}

int ddh_pencil_solve_recombined_tiled(ddh_handle pack, int lu_id, int nterms, const double *const *xs_h,
                                      const double *alpha_h, int p_mat_id, double *work, double *x,
                                      const unsigned char *zero_rows, const unsigned char *skip_rows, void *stream) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (lu_id < 0 || lu_id >= (int)pp->lus.size()) return fail("pencil_solve: bad LU id");
    if (pp->dev.nf != 2 || (pp->dev.nx & 7) || (pp->dev.ny & 7))
        return fail("pencil_solve_recombined_tiled: two Fourier axes with storage sizes that are multiples of 8");
    {
        int uf, cbv;
        choose_variant<2>(pp, pp->lus[lu_id]->dev, uf, cbv);
        if (pp->lus[lu_id]->dev.pair) uf = 0;
        if (uf || !lean_forward_ok(pp->lus[lu_id]->dev))
            return fail("pencil_solve_recombined_tiled: this factorization does not run the lean forward sweep, the only one "
                        "that reads tile-major terms (ddh_pencil_lu_info)");
    }
    return solve_recombined_impl(pack, lu_id, nterms, xs_h, alpha_h, p_mat_id, work, x, zero_rows, skip_rows, 1, stream);
}

static int solve_recombined_impl(ddh_handle pack, int lu_id, int nterms, const double *const *xs_h, const double *alpha_h,
                                 int p_mat_id, double *work, double *x, const unsigned char *zero_rows,
                                 const unsigned char *skip_rows, int terms_tiled, void *stream) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (lu_id < 0 || lu_id >= (int)pp->lus.size()) return fail("pencil_solve: bad LU id");
    if (p_mat_id < 0 || p_mat_id >= (int)pp->mats.size()) return fail("pencil_solve_recombined: bad matrix id");
    if (nterms < 1 || nterms > RHS_MAX) return fail("pencil_solve_recombined: 1 to 8 right-hand-side terms");
    if (work == x || !work) return fail("pencil_solve_recombined: work and x must be distinct buffers");
    RhsSrc r;
    memset(&r, 0, sizeof(r));
    r.n = nterms;
    for (int t = 0; t < nterms; ++t) {
        if (xs_h[t] == x || xs_h[t] == work) return fail("pencil_solve: in-place unsupported");
        r.p[t] = xs_h[t];
        r.a[t] = alpha_h[t];
    }
    r.zrow = zero_rows;
    r.skip = skip_rows;
    r.tiled = terms_tiled;
    LuFactor *lu = pp->lus[lu_id];
    hipStream_t s = as_stream(stream);
    static const bool no_fuse = getenv("DDH_NO_PFUSE") != nullptr;
    bool did = false;
    int st;
    if (pp->dev.nf == 2 && !no_fuse) {
        if ((st = build_pband(pp, lu, p_mat_id))) return st;
        // the fused kernel writes x directly; if this launch cannot fuse, the sweeps must write the work vector instead:
        // decide first (same logic as launch_solve) by a dry query
        int uf, cbv;
        choose_variant<2>(pp, lu->dev, uf, cbv);
        if (lu->dev.pband != nullptr && lu->dev.real && !cbv && lu->dev.W <= 48 && lu->dev.n > 0) {
            st = launch_solve<2>(pp, lu, r, x, s, true, &did);
            if (st) return st;
            if (did) return 0;
            return fail("pencil_solve_recombined: internal error (fused variant not taken)");
        }
    }
    r.skip = nullptr;       // (the unfused path recombines from `work` with a mat-vec: every row of it must be written)
    if (pp->dev.nf == 1 && lu->d_binv && lu->dev.real && lu->dev.n == lu->dev.nsplit * lu->dev.nh) {
        const LuDev &d = lu->dev;
        const unsigned nt = (unsigned)std::min(1024, (d.nh + 63) / 64 * 64);
        hipLaunchKernelGGL(blockinv_solve_kernel, dim3((unsigned)(pp->dev.ncells * d.nsplit)), dim3(nt), (size_t)d.nh * sizeof(double2),
                           s, pp->dev, d, r, lu->d_binv, work);
        DDH_HIP(hipGetLastError());
        st = finish_solve<1>(pp, lu, r, work, s, 0);
    } else if (pp->dev.nf == 2) st = launch_solve<2>(pp, lu, r, work, s);
    else if (pp->dev.nf == 1) st = launch_solve<1>(pp, lu, r, work, s);
    else st = launch_solve<0>(pp, lu, r, work, s);
    if (st) return st;
    PostSolve none;
    memset(&none, 0, sizeof(none));
    // (a tile-major state, ddh_pencil_set_state_tiled: `work` was written tile-major by the sweeps, and x must be too)
    if (pp->dev.xtile == 2) return fail("pencil_solve_recombined: a kx-band-major state needs the fused recombination");
    return launch_matvec(pp, p_mat_id, work, x, none, stream, 0, pp->dev.xtile);
}

int ddh_pencil_lu_row_widths(ddh_handle pack, int lu_id, int *wrow_h) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (lu_id < 0 || lu_id >= (int)pp->lus.size()) return fail("pencil_lu_row_widths: bad LU id");
    const LuDev &d = pp->lus[lu_id]->dev;
    if (d.n <= 0) return 0;
    int *dw = nullptr;
    DDH_HIP(hipMalloc((void **)&dw, d.n * sizeof(int)));
    DDH_HIP(hipMemset(dw, 0, d.n * sizeof(int)));
    const unsigned blocks = (unsigned)((d.GL + 255) / 256);
    const long only = getenv("DDH_LUW_BLOCK") ? atol(getenv("DDH_LUW_BLOCK")) : -1;      // (diagnostic: one block of 64)
    const dim3 wgrid(blocks, (unsigned)((d.n + LUW_ROWS - 1) / LUW_ROWS));
    if (d.real) hipLaunchKernelGGL(lu_width_kernel<true>, wgrid, dim3(256), 0, 0, d, dw, only);
    else hipLaunchKernelGGL(lu_width_kernel<false>, wgrid, dim3(256), 0, 0, d, dw, only);
    DDH_HIP(hipGetLastError());
    DDH_HIP(hipMemcpy(wrow_h, dw, d.n * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(dw);
    return 0;
}

int ddh_pencil_lu_info(ddh_handle pack, int lu_id, int *info_h) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (lu_id < 0 || lu_id >= (int)pp->lus.size()) return fail("pencil_lu_info: bad LU id");
    const LuDev &d = pp->lus[lu_id]->dev;
    int use_fwd = 0, cb = 0;
    switch (pp->dev.nf) {
        case 0: choose_variant<0>(pp, d, use_fwd, cb); break;
        case 1: choose_variant<1>(pp, d, use_fwd, cb); break;
        default: choose_variant<2>(pp, d, use_fwd, cb); break;
    }
    if (d.pair) use_fwd = cb = 0;
    const bool lean = pp->dev.nf == 2 && !use_fwd && lean_forward_ok(d);
    const int v[12] = {d.n, d.nb, d.kl, d.ku, d.W, d.BW, d.nsplit, d.nh, use_fwd ? 2 : (lean ? 1 : 0), cb, d.pair, d.real};
    for (int i = 0; i < 12; ++i) info_h[i] = v[i];
    return 0;
}

int ddh_pencil_lu_bytes(ddh_handle pack, int lu_id, size_t *bytes) {
    PencilPack *pp = (PencilPack *)lookup_handle(pack, H_PENCIL);
    if (!pp) return -1;
    if (lu_id < 0 || lu_id >= (int)pp->lus.size()) return fail("pencil_lu_bytes: bad LU id");
    *bytes = pp->lus[lu_id]->bytes;
    return 0;
}

}  // extern "C"
