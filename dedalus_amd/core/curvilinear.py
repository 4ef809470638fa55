"""
Curvilinear transform plans on the device.

SWSHColatitudeTransform mirrors the reference's plan class of the same name
(core/transforms.py:1251-1340): built from (Ntheta, Lmax, m_maps, s), it maps between the reduced 4-D
views gdata[n0, m-axis, Ntheta, n3] and cdata[n0, m-axis, ell-axis, n3].  The matrices are built on the
host by dedalus_amd/tools/sphere.py; the per-m loop of small matmuls is ONE grouped launch
(ddh_grouped_mmt_forward / backward, csrc/ddh_swsh.hip).
"""

import numpy as np

from ..tools import sphere


def m_maps_to_groups(m_maps, Lmax):
    """m_maps entries (m, mg_slice, mc_slice, ell_slice) (SphereBasis.m_maps, core/basis.py:2939-2970)
    -> integer rows (m, g_start, c_start, count, ell_start, ell_step, n_ell)."""
    rows = []
    for (m, mg, mc, es) in m_maps:
        n_ell = Lmax + 1 - abs(m) if abs(m) <= Lmax else 0
        step = -1 if es.step == -1 else 1
        rows.append((int(m), int(mg.start), int(mc.start), int(mg.stop - mg.start), int(es.start), step, n_ell))
    return np.array(rows, dtype=np.int64).reshape(-1, 7)


class SWSHColatitudeTransform:
    """Spin-weighted spherical harmonic transform along colatitude for spin weight s."""

    def __init__(self, Ntheta, Lmax, m_maps, s, executor=None):
        self.Ntheta, self.Lmax, self.s = int(Ntheta), int(Lmax), int(s)
        self.groups = m_maps if isinstance(m_maps, np.ndarray) else m_maps_to_groups(m_maps, Lmax)
        if executor is None:
            from ..executor import HipExecutor
            executor = HipExecutor()
        self.ex = executor
        ms = []
        for row in self.groups:
            m = int(row[0])
            if abs(m) <= self.Lmax and m not in ms:
                ms.append(m)
        mats = {m: sphere.swsh_matrices(self.Ntheta, self.Lmax, m, self.s) for m in ms}
        self.plan = executor.make_grouped_mmt(self.Ntheta, self.groups, ms, [mats[m][0] for m in ms],
                                              [mats[m][1] for m in ms])

    def forward_reduced(self, gdata, cdata):
        """gdata [n0, n1g, Ntheta, n3] -> cdata [n0, n1c, n2c, n3] (groups with |m| > Lmax are left untouched)."""
        self.plan.forward(gdata, cdata)

    def backward_reduced(self, cdata, gdata):
        """cdata -> gdata (groups with |m| > Lmax are zero-filled)."""
        self.plan.backward(cdata, gdata)


def ell_maps_to_rows(ell_maps):
    """ell_maps entries (ell, m_slice, ell_slice) (ShellBasis.ell_maps) -> integer rows (ell, m0, m1, l0, l1)."""
    return np.array([(int(e), int(ms.start), int(ms.stop), int(ls.start), int(ls.stop)) for (e, ms, ls) in ell_maps],
                    dtype=np.int64).reshape(-1, 5)


def recombination_tables(ell_rows, shape12, rank):
    """(slot_map [n1][n2] int32, forward matrices, backward matrices) for the regularity recombination of a
    rank-`rank` tensor: per slot the ordered product of Q(ell)^T (forward) / Q(ell) (backward) over all
    ell_maps entries covering the slot, exactly as the reference's sequential loop applies them
    (core/basis.py:3595-3626; the entries are bounding boxes and can overlap)."""
    n1, n2 = shape12
    seqs = [[[] for _ in range(n2)] for _ in range(n1)]
    for (ell, m0, m1, l0, l1) in ell_rows:
        for i1 in range(m0, m1):
            for i2 in range(l0, l1):
                seqs[i1][i2].append(int(ell))
    Q = {}
    index, fwd, bwd = {}, [], []
    slot = -np.ones((n1, n2), dtype=np.int32)
    nc = 3 ** rank
    for i1 in range(n1):
        for i2 in range(n2):
            key = tuple(seqs[i1][i2])
            if not key:
                continue
            if key not in index:
                F, B = np.eye(nc), np.eye(nc)
                for ell in key:
                    if ell not in Q:
                        Q[ell] = sphere.intertwiner(ell, rank)
                    F = Q[ell].T @ F
                    B = Q[ell] @ B
                index[key] = len(fwd)
                fwd.append(F)
                bwd.append(B)
            slot[i1, i2] = index[key]
    return slot, np.array(fwd).reshape(-1, nc, nc), np.array(bwd).reshape(-1, nc, nc)


class RegularityRecombination:
    """forward/backward_regularity_recombination + radial factor of a shell tensor field on the device
    (core/basis.py:3595-3626, 4474-4508): one launch of ddh_regularity_recombine each way."""

    def __init__(self, ell_rows, shape12, rank, executor=None):
        if executor is None:
            from ..executor import HipExecutor
            executor = HipExecutor()
        self.ex, self.rank = executor, rank
        slot, fwd, bwd = recombination_tables(ell_rows, shape12, rank)
        self.fwd = executor.make_recombination(slot, fwd) if rank > 0 else None
        self.bwd = executor.make_recombination(slot, bwd) if rank > 0 else None

    def forward(self, data, radial_factor=None):
        self.ex.regularity_recombine(data, self.fwd, radial_factor)

    def backward(self, data, radial_factor=None):
        self.ex.regularity_recombine(data, self.bwd, radial_factor)
