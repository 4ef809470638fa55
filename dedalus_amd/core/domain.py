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

    def global_coeff_shape(self):
        return tuple(1 if b is None else b.coeff_size for b in self.by_axis)

    def global_grid_shape(self, scales):
        return tuple(1 if b is None else b.grid_size(s) for b, s in zip(self.by_axis, scales))

    def coeff_shape(self):
        """LOCAL user-axis-order coefficient shape (kx-sharded on several ranks)."""
        shape = list(self.global_coeff_shape())
        ax = self.dist.shard_coeff_axis
        if ax is not None and self.by_axis[ax] is not None:
            lo, hi = self.dist.local_block(shape[ax] // 2)      # blocks of whole (cos, msin) pairs
            shape[ax] = 2 * (hi - lo)
        return tuple(shape)

    def grid_shape(self, scales):
        """LOCAL grid shape (sharded along the Jacobi axis on several ranks)."""
        shape = list(self.global_grid_shape(scales))
        ax = self.dist.shard_grid_axis
        if ax is not None and self.by_axis[ax] is not None:
            lo, hi = self.dist.local_block(shape[ax])
            shape[ax] = hi - lo
        return tuple(shape)

    def local_slices(self, layout, scales=None):
        """Slices of the global array (user axis order) owned by this rank."""
        sl = [slice(None)] * self.dist.dim
        if layout == "c":
            ax = self.dist.shard_coeff_axis
            if ax is not None and self.by_axis[ax] is not None:
                lo, hi = self.dist.local_block(self.by_axis[ax].coeff_size // 2)
                sl[ax] = slice(2 * lo, 2 * hi)
        else:
            ax = self.dist.shard_grid_axis
            if ax is not None and self.by_axis[ax] is not None:
                lo, hi = self.dist.local_block(self.by_axis[ax].grid_size(scales[ax]))
                sl[ax] = slice(lo, hi)
        return tuple(sl)

    # ---- internal ("z-major") storage order -----------------------------------------------------
    def storage_coeff_shape(self):
        cs = self.coeff_shape()
        return tuple(cs[ax] for ax in self.dist.storage_order)

    def storage_grid_shape(self, scales):
        gs = self.grid_shape(scales)
        return tuple(gs[ax] for ax in self.dist.storage_order)
