"""Product-of-bases bookkeeping (the role of dedalus/core/domain.py:1-227, metadata only)."""

import numpy as np


class Domain:
    def __init__(self, dist, bases):
        self.dist = dist
        by_axis = [None] * dist.dim
        for b in bases:
            if b is None:
                continue
            ax = dist.coord_axis(b.coord)
            if by_axis[ax] is not None and by_axis[ax] != b:
                raise ValueError("Overlapping bases specified.")
            by_axis[ax] = b
        self.by_axis = tuple(by_axis)
        self.bases = tuple(b for b in by_axis if b is not None)

    def __eq__(self, other):
        return isinstance(other, Domain) and self.by_axis == other.by_axis

    def __hash__(self):
        return hash(self.by_axis)

    def get_basis(self, coord):
        return self.by_axis[self.dist.coord_axis(coord)]

    def replace(self, axis, basis):
        lst = list(self.by_axis)
        lst[axis] = basis
        return Domain(self.dist, lst)

    def combine(self, other, op):
        out = []
        for a, b in zip(self.by_axis, other.by_axis):
            if a is None:
                out.append(b)
            elif b is None:
                out.append(a)
            else:
                r = a + b if op == "add" else a * b
                if r is NotImplemented:
                    raise ValueError("incompatible bases %r and %r" % (a, b))
                out.append(r)
        return Domain(self.dist, out)

    @property
    def dealias(self):
        return tuple(1.0 if b is None else b.dealias for b in self.by_axis)

    def coeff_shape(self):
        """User-axis-order coefficient shape."""
        return tuple(1 if b is None else b.coeff_size for b in self.by_axis)

    def grid_shape(self, scales):
        return tuple(1 if b is None else b.grid_size(s) for b, s in zip(self.by_axis, scales))

    # ---- internal ("z-major") storage order -----------------------------------------------------
    def storage_coeff_shape(self):
        cs = self.coeff_shape()
        return tuple(cs[ax] for ax in self.dist.storage_order)

    def storage_grid_shape(self, scales):
        gs = self.grid_shape(scales)
        return tuple(gs[ax] for ax in self.dist.storage_order)
