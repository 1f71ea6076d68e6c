"""The geometry kernel's range-tested sqrt / division fast paths (csrc/fpexact.hpp) give the bits of the IEEE operators:
random, special and range-edge arguments, compared on the device (smgpu_debug_selftest_fpexact)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 20261003, 0xDEADBEEF])
def test_fast_sqrt_and_division_match_the_ieee_operators(seed):
    from smoothmesh_amd import _ffi

    lib = _ffi.lib()
    bad = C.c_int64(-1)
    rc = lib.smgpu_debug_selftest_fpexact(0, seed, 40_000_000, C.byref(bad))
    assert rc == 0, lib.smgpu_last_error().decode()
    assert bad.value == 0
