"""The threaded C++ generators (bench-size volumes) produce exactly the numpy definitions."""
import numpy as np
import pytest

from volumerendering_amd import synth


@pytest.mark.parametrize("n", [8, 31, 48])
def test_fast_generators_match_numpy_definitions(n):
    assert np.array_equal(synth.ct_phantom_raw_fast(n), synth.ct_phantom_raw(n))
    assert np.array_equal(synth.sphere_raw_fast(n), synth.sphere_raw(n))
    assert np.array_equal(synth.mask_vec4_fast(n), synth.mask_vec4(n))


def test_phantom_has_the_three_tissue_classes():
    v = synth.ct_phantom_raw(48)
    assert v.max() <= 4095 and (v == 0).mean() > 0.3
    assert ((v > 900) & (v < 1200)).mean() > 0.1 and (v > 2400).mean() > 0.02
    assert np.array_equal(synth.ct_phantom_raw(48, 10, 20), v[10:20])
