"""CPU: the host logic of the tile-major state vector (round 6) -- SystemBuffer's natural shadows and their validity, the
two stored orders (tile-major rows, kx-band-major) -- against the documented permutations, with a stand-in executor that
implements ddh_tile_rows in numpy on torch CPU tensors.  The kernels themselves: tests/test_gpu_state_tiling.py."""
import numpy as np
import pytest
import torch

from dedalus_amd.core.evaluator import SystemBuffer


def _tile(a):
    R, nx, ny = a.shape
    return np.ascontiguousarray(a.reshape(R, nx // 8, 8, ny // 8, 8).transpose(0, 1, 3, 2, 4)).reshape(R, nx, ny)


def _band(a):
    R, nx, ny = a.shape
    return np.ascontiguousarray(a.reshape(R, nx // 8, 8, ny // 8, 8).transpose(1, 0, 3, 2, 4)).reshape(R, nx, ny)


class FakeEx:
    """ddh_tile_rows restated: the tiled side is addressed from its first element, like the library does."""

    def __init__(self):
        self.calls = []

    def empty(self, shape):
        return torch.full(tuple(shape), float("nan"), dtype=torch.float64)

    def tile_rows(self, src, dst, nrows, nx, ny, to_tiled, band_rows=0):
        self.calls.append((nrows, bool(to_tiled), band_rows))
        nat, til = (src, dst) if to_tiled else (dst, src)
        # flat views starting at the first element of each side (the tiled side may be a strided view of a larger vector)
        tf = torch.as_strided(til, (til.untyped_storage().size() // 8 - til.storage_offset(),), (1,), til.storage_offset())
        for r in range(nrows):
            for kxb in range(nx // 8):
                blk = nat[r, 8 * kxb:8 * kxb + 8, :].reshape(8, ny // 8, 8).permute(1, 0, 2).reshape(-1)      # [ky/8][8][8]
                off = ((kxb * band_rows + r) * 8 * ny) if band_rows else (r * nx * ny + kxb * 8 * ny)
                if to_tiled:
                    tf[off:off + 8 * ny] = blk
                else:
                    nat[r, 8 * kxb:8 * kxb + 8, :] = tf[off:off + 8 * ny].reshape(ny // 8, 8, 8).permute(1, 0, 2).reshape(8, ny)


@pytest.mark.parametrize("banded", [False, True])
def test_natural_shadows_follow_the_tiled_state(banded):
    rng = np.random.default_rng(3)
    R, nx, ny = 7, 16, 24
    x = rng.standard_normal((R, nx, ny))
    ex = FakeEx()
    sb = SystemBuffer(torch.from_numpy((_band if banded else _tile)(x).copy()), R)
    sb.tiled, sb.banded = ny, (R if banded else 0)
    sb.ranges = [(0, 3), (3, 2), (5, 2)]
    sb.valid = {0: False, 3: False, 5: False}
    # a variable's rows, converted on first use only
    rows = sb.natural_rows(ex, 3, 2)
    assert np.array_equal(rows.numpy(), x[3:5]) and sb.valid == {0: False, 3: True, 5: False}
    n = len(ex.calls)
    sb.natural_rows(ex, 3, 2)
    assert len(ex.calls) == n                                   # valid: no second conversion
    # the whole state
    assert np.array_equal(sb.natural(ex).numpy(), x)
    # the solver writes the state: every shadow is stale
    sb.invalidate()
    assert not any(sb.valid.values())
    # the user rewrites a variable in the natural order: committed into the tiled state, the other rows untouched
    new = rng.standard_normal((2, nx, ny))
    sb.natural_rows(ex, 5, 2, current=False)[...] = torch.from_numpy(new)
    sb.commit_rows(ex, 5, 2)
    x[5:7] = new
    assert np.array_equal(sb.array.numpy(), (_band if banded else _tile)(x))
    assert sb.valid[5] and not sb.valid[0]
    # the stored rows of a variable start where the kernels expect them
    v = sb.state_rows(3, 2)
    assert v.data_ptr() == sb.array.data_ptr() + 8 * (3 * 8 * ny if banded else 3 * nx * ny)
    assert v.numel() == 2 * nx * ny


def test_untiled_buffer_is_its_own_natural_form():
    a = torch.zeros((2, 8, 8), dtype=torch.float64)
    sb = SystemBuffer(a, 2)
    assert sb.natural(FakeEx()) is a
