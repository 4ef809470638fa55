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
