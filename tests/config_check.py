"""Line up this package's per-m / per-ell system matrices with the REFERENCE's own subproblem matrices
(tests/golden/config_sphere.npz, config_shell.npz, made by oracle/make_golden_config.py from core/subsystems.py:497-596).

Both sides are tagged the same way: every variable's (equation field's) coefficient array -- the reference layout at
the user boundary -- is filled with TAG * index + flat position.  The reference recorded which tag each unknown /
equation of a subproblem carries; here the tags are pushed through this package's own layout conversion into the
solver's internal vectors, so each internal slot knows which reference unknown it is.  Nothing about either layout is
assumed beyond "field['c'] means the same array on both sides" (tested separately on small problems)."""
import numpy as np

TAG = 2 ** 40


def _tags(i, shape):
    # + 1: a slot that holds 0 afterwards was never written (it has no counterpart in the reference's layout)
    return (i * TAG + 1 + np.arange(int(np.prod(shape)), dtype=np.float64)).reshape(shape)


def _split(t):
    """tag array -> (sign, field index, flat position, tagged?)"""
    sign = np.where(t < 0, -1.0, 1.0)
    a = np.rint(np.abs(t)).astype(np.int64)
    assert np.all(a == np.abs(t)), "the layout conversion is not a signed permutation"
    has = a > 0
    a = np.where(has, a - 1, 0)
    return sign, a // TAG, a % TAG, has


def csr(gold, tag):
    from scipy import sparse
    shape = tuple(int(x) for x in gold[tag + "_shape"])
    return sparse.csr_matrix((gold[tag + "_data"], gold[tag + "_indices"], gold[tag + "_indptr"]), shape=shape)


def internal_tags(solver, make_eq_field, place):
    """-> (Tin, Tout): the solver's state vector and an equation-space vector filled with reference tags (host arrays in
    the solver's internal layout).  make_eq_field(eq) -> a field with the equation's tensor signature and bases;
    place(Tout, j, c) writes equation j's internal coefficient array c into the system vector Tout."""
    ex = solver.ex
    saved = []
    for i, v in enumerate(solver.variables):
        c = np.array(v['c'])
        saved.append(c)
        v['c'] = _tags(i, c.shape)
    solver.sync_state_to_device()
    Tin = np.array(ex.download(solver.state_natural() if hasattr(solver, "state_natural") else solver.X))
    for v, c in zip(solver.variables, saved):
        v['c'] = c
    solver.sync_state_to_device()
    Tout = np.zeros_like(Tin)
    for j, eq in enumerate(solver.problem.equations):
        f = make_eq_field(eq)
        if f is None:
            continue
        c = np.array(f['c'])
        f['c'] = _tags(j, c.shape)
        if hasattr(f, "require_coeff_space"):
            place(Tout, j, np.array(ex.download(f.require_coeff_space())))
        else:                                            # a constant (the gauge equation): one number at m = 0, ell = 0
            place(Tout, j, float(np.asarray(f['c']).reshape(-1)[0]))
    return Tin, Tout


def compare_group(A_mine, slots_in, slots_out, Tin, Tout, gold, tag, a, b, col=0, ignore=(), want_maps=False):
    """A_mine: real matrix on this package's internal real slots of one group; slots_*: index arrays into Tin / Tout
    (flattened) of those slots, in A_mine's order.  -> max |A_ref - A_mine (reference order)| / max |A_ref|."""
    A_ref = (a * csr(gold, tag + "M_min") + b * csr(gold, tag + "L_min")).toarray()
    want_in = gold[tag + "in_var"][:, col].astype(np.int64) * TAG + gold[tag + "in_flat"][:, col]
    want_out = gold[tag + "out_eq"][:, col].astype(np.int64) * TAG + gold[tag + "out_flat"][:, col]
    si, fi, pi, hi = _split(Tin.reshape(-1)[slots_in])
    so, fo, po, ho = _split(Tout.reshape(-1)[slots_out])
    have_in = {int(k): n for n, k in enumerate(fi * TAG + pi) if hi[n]}
    have_out = {int(k): n for n, k in enumerate(fo * TAG + po) if ho[n]}
    ci = np.array([have_in[int(k)] for k in want_in])
    ro = np.array([have_out[int(k)] for k in want_out])
    mine = (so[ro][:, None] * A_mine[np.ix_(ro, ci)]) * si[ci][None, :]
    # everything outside the reference's valid block must be empty on this side too
    # coefficients of the reference layout that are not in its subproblem (invalid modes) must be empty here as well;
    # slots without a counterpart (the unused imaginary part of m = 0) are not looked at
    ign = np.asarray(ignore, dtype=np.int64)
    live_r, live_c = np.setdiff1d(np.nonzero(ho)[0], ign), np.setdiff1d(np.nonzero(hi)[0], ign)
    rest_r, rest_c = np.setdiff1d(live_r, ro), np.setdiff1d(live_c, ci)
    assert not np.any(A_mine[np.ix_(rest_r, live_c)]) and not np.any(A_mine[np.ix_(live_r, rest_c)]), \
        "entries outside the reference's valid modes"
    err = float(np.abs(A_ref - mine).max() / np.abs(A_ref).max())
    if not want_maps:
        return err, A_ref.shape[0]
    # maps into the flattened internal vectors: reference unknown k <-> Tin.flat[cols[k]] * csign[k], equation k likewise
    return err, dict(A_ref=A_ref, cols=np.asarray(slots_in)[ci], csign=si[ci], rows=np.asarray(slots_out)[ro], rsign=so[ro])


# ---- sphere: complex per-m matrices on z = cos + i msin ------------------------------------------------------------------

def sphere_group(solver, m, a, b):
    """real (2 n x 2 n) matrix of (a M + b L) at wavenumber m on the internal slots [comp][2 m + part][ell >= m], and the
    flat indices of those slots in X"""
    R, nl = solver.R, solver.basis.nl
    ne = nl - m
    Ac = a * solver._dense(solver.M_tl, m) + b * solver._dense(solver.L_tl, m)
    n = R * ne
    A = np.zeros((2 * n, 2 * n))
    A[:n, :n], A[:n, n:], A[n:, :n], A[n:, n:] = Ac.real, -Ac.imag, Ac.imag, Ac.real     # (C, S) = A (c, s)
    idx = np.arange(R * 2 * solver.basis.nm * nl).reshape(R, 2 * solver.basis.nm, nl)
    slots = np.concatenate([idx[:, 2 * m + part, m:].reshape(-1) for part in (0, 1)])
    # m = 0: the imaginary halves are no modes (always-zero data the complex arithmetic carries along)
    ignore = np.arange(n, 2 * n) if m == 0 else np.zeros(0, dtype=np.int64)
    return A, slots, ignore


def sphere_tags(solver, d3):
    coords, basis, dist = solver.basis.coordsys, solver.basis, solver.dist

    def make(eq):
        if eq.get("constant"):
            return None
        return dist.Field(bases=basis) if eq["rank"] == 0 else dist.TensorField(coords, bases=basis, order=eq["rank"])

    def place(T, j, c):
        r0 = solver.row0[j]
        T[r0:r0 + c.shape[0]] = c

    return internal_tags(solver, make, place)


# ---- shell: real per-ell matrices, every (m, part) slot of an ell is one right-hand side -------------------------------------

def shell_tags(solver, d3):
    from dedalus_amd.core.shell import ShellBasis, SurfaceBasis
    dist, shell = solver.dist, solver.shell
    coords = shell.coordsys

    def make(eq):
        basis, rank = eq["basis"], eq["rank"]
        bases = () if basis is None else (basis,)
        if rank == 0:
            return dist.Field(bases=basis) if basis is not None else dist.Field()
        return dist.TensorField(coords, bases=basis, order=rank) if rank > 1 else dist.VectorField(coords, bases=basis)

    def place(T, j, c):
        T4 = T.reshape(solver.R, 2 * solver.nm, solver.nl, solver.Nr)
        if isinstance(c, float):
            sc, off, nr = solver.emap[j][0]
            T4[sc, 0, 0, off] = c
            return
        for comp, (sc, off, nr) in enumerate(solver.emap[j]):
            T4[sc, :, :, off:off + nr] = c[comp][..., :nr]

    return internal_tags(solver, make, place)


def shell_group(solver, ell, a, b, Tin, gold, tag, col=0):
    """real matrix of (a M + b L) at this ell on the internal slots [part][system component][n] of the azimuthal
    wavenumber m the reference's subsystem `col` of that subproblem belongs to (the reference keeps the cos and msin
    parts of a wavenumber in one vector: the same block twice), the flat indices of those slots, m, and the slots to
    leave out of the emptiness check (m = 0: the msin half is no mode)"""
    A1 = a * solver._dense(solver.M_tl, ell) + b * solver._dense(solver.L_tl, ell)
    n = A1.shape[0]
    A = np.zeros((2 * n, 2 * n))
    A[:n, :n] = A1
    A[n:, n:] = A1
    R, nm2, nl, Nr = solver.R, 2 * solver.nm, solver.nl, solver.Nr
    # which wavenumber: look one of the wanted tags up in the whole vector
    want = int(gold[tag + "in_var"][0, col]) * TAG + int(gold[tag + "in_flat"][0, col]) + 1
    pos = np.nonzero(np.abs(Tin.reshape(-1)) == float(want))[0]
    assert pos.size == 1, (ell, col, pos.size)
    m = int(np.unravel_index(pos[0], (R, nm2, nl, Nr))[1]) // 2
    idx = np.arange(R * nm2 * nl * Nr).reshape(R, nm2, nl, Nr)
    slots = np.concatenate([idx[:, 2 * m + part, ell, :].reshape(-1) for part in (0, 1)])
    ignore = np.arange(n, 2 * n) if m == 0 else np.zeros(0, dtype=np.int64)
    return A, slots, m, ignore


def check_solve(maps, rhs, x):
    """One recorded solve (internal system vectors) against the REFERENCE's matrix of the group, in the reference's
    order: -> (residual |A x - r| / |r|, error |x - LAPACK(A, r)| / |x|)"""
    A = maps["A_ref"]
    r = rhs.reshape(-1)[maps["rows"]] * maps["rsign"]
    xr = x.reshape(-1)[maps["cols"]] * maps["csign"]
    nr = np.linalg.norm(r)
    if nr == 0.0:
        return 0.0, float(np.linalg.norm(xr))
    ref = np.linalg.solve(A, r)
    return float(np.linalg.norm(A @ xr - r) / nr), float(np.linalg.norm(xr - ref) / max(np.linalg.norm(ref), 1e-300))
