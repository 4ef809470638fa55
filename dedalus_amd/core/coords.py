"""Cartesian coordinate systems (S2Coordinates: core/sphere.py, SphericalCoordinates: core/shell.py). Mirrors the names of dedalus/core/coords.py:1-413."""

import numpy as np


class Coordinate:
    dim = 1

    def __init__(self, name, cs=None):
        self.name = name
        self.cs = cs
        self.coords = (self,)

    def __repr__(self):
        return "Coordinate(%s)" % self.name

    def check_bounds(self, bounds):
        if len(bounds) != 2 or not bounds[1] > bounds[0]:
            raise ValueError("bounds must be (lower, upper) with upper > lower")


class CartesianCoordinates:
    def __init__(self, *names):
        if len(set(names)) < len(names):
            raise ValueError("Must specify unique names.")
        self.names = names
        self.dim = len(names)
        self.coords = tuple(Coordinate(n, cs=self) for n in names)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.coords[self.names.index(key)]
        return self.coords[key]

    def __iter__(self):
        return iter(self.coords)

    def index(self, coord):
        return self.coords.index(coord)

    def unit_vector_fields(self, dist):
        """Constant vector fields e_i (coords.py:unit_vector_fields)."""
        fields = []
        for i, c in enumerate(self.coords):
            f = dist.VectorField(self, name="e" + c.name)
            data = np.zeros((self.dim,) + (1,) * dist.dim)
            data[i] = 1.0
            f["g"] = data
            fields.append(f)
        return tuple(fields)
