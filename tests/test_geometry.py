"""Geometry sampling is host numpy and must be BIT-EXACT with the reference.

* tests/golden/geometry.npz was produced by the reference's own geometry code
  (tests/golden/make_geometry_golden.py); every array must match exactly, dtype included.
* The doctest known answers of /root/reference/ppsci/geometry/geometry.py:157-183, 257-281, 361-376
  (np.random.seed(42)) are checked literally as well."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import geometry_cases  # noqa: E402

from paddlescience_amd import geometry as G  # noqa: E402


def test_matches_reference_generated_fixtures():
    gold = np.load(os.path.join(HERE, "golden", "geometry.npz"))
    mine = geometry_cases.run(G)
    assert set(mine) == set(gold.files)
    for k in gold.files:
        a, b = mine[k], gold[k]
        assert a.dtype == b.dtype, (k, a.dtype, b.dtype)
        assert a.shape == b.shape, (k, a.shape, b.shape)
        assert np.array_equal(a, b), k


def test_reference_doctest_known_answers():
    np.random.seed(42)
    iv = G.Interval(0, 1).sample_interior(2)
    np.testing.assert_array_equal(iv["x"], np.array([[0.37454012], [0.9507143]], dtype=np.float32))
    np.testing.assert_array_equal(iv["sdf"], np.array([[0.37454012], [0.04928571]], dtype=np.float32))
    r = G.Rectangle((0, 0), (1, 1)).sample_interior(2, "pseudo", None, False, True)
    np.testing.assert_array_equal(r["x"], np.array([[0.7319939], [0.15601864]], dtype=np.float32))
    np.testing.assert_array_equal(r["y"], np.array([[0.5986585], [0.15599452]], dtype=np.float32))
    np.testing.assert_array_equal(r["sdf"], np.array([[0.2680061], [0.15599453]], dtype=np.float32))
    np.testing.assert_allclose(r["sdf__x"], np.array([[-1.0001659], [0.25868416]], dtype=np.float32), rtol=1e-6)
    np.testing.assert_allclose(r["sdf__y"], np.array([[-0.0], [0.74118376]], dtype=np.float32), rtol=1e-6)
    c = G.Cuboid((0, 0, 0), (1, 1, 1)).sample_interior(2, "pseudo", None, True, True)
    np.testing.assert_array_equal(c["z"], np.array([[0.0], [1.0]], dtype=np.float32))
    np.testing.assert_allclose(c["sdf__z"], np.array([[0.50008297], [-0.49948692]], dtype=np.float32), rtol=1e-6)

    np.random.seed(42)
    b = G.Interval(0, 1).sample_boundary(2)
    np.testing.assert_array_equal(b["x"], np.array([[0.0], [1.0]], dtype=np.float32))
    np.testing.assert_array_equal(b["normal_x"], np.array([[-1.0], [1.0]], dtype=np.float32))
    rb = G.Rectangle((0, 0), (1, 1)).sample_boundary(2)
    np.testing.assert_array_equal(rb["x"], np.array([[1.0], [0.0]], dtype=np.float32))
    np.testing.assert_array_equal(rb["y"], np.array([[0.49816048], [0.19714284]], dtype=np.float32))
    np.testing.assert_array_equal(rb["normal_x"], np.array([[1.0], [-1.0]], dtype=np.float32))
    cb = G.Cuboid((0, 0, 0), (1, 1, 1)).sample_boundary(2)
    np.testing.assert_array_equal(cb["x"], np.array([[0.83244264], [0.18182497]], dtype=np.float32))
    np.testing.assert_array_equal(cb["z"], np.array([[0.0], [1.0]], dtype=np.float32))


def test_grids_of_the_baseline_configs():
    """Laplace 101x101 on [0,1]^2 and LDC 99x99 on [-0.05,0.05]^2."""
    x = G.Rectangle((0.0, 0.0), (1.0, 1.0)).uniform_points(10201)
    assert x.shape == (10201, 2) and len(np.unique(x[:, 0])) == 101
    y = G.Rectangle((-0.05, -0.05), (0.05, 0.05)).uniform_points(9801)
    assert y.shape == (9801, 2) and len(np.unique(y[:, 1])) == 99


def test_errors():
    with pytest.raises(ValueError):
        G.Hypercube((0, 0), (1,))
    with pytest.raises(ValueError):
        G.Rectangle((1, 0), (0, 1))
    with pytest.raises(ValueError):
        G.TimeXGeometry(G.TimeDomain(0, 1), G.Interval(0, 1)).random_points(5)
    with pytest.raises(ValueError):
        G.Rectangle((0, 0), (1, 1)).sample_interior(5, criteria=lambda x, y: x > 2)
    with pytest.raises(ValueError):
        G.Rectangle((0, 0), (1, 1)).sample_interior(5, random="NoSuchSampler")


@pytest.mark.parametrize("method", ["LHS", "Halton", "Hammersley", "Sobol"])
@pytest.mark.parametrize("ndim", [1, 2, 3])
def test_quasi_random_samplers(method, ndim):
    """sampler.py:60-92: shape / dtype / open unit cube, the reference's skip rules (no all-zero point, Sobol also
    without [0.5, ...]), low discrepancy (every axis-aligned half holds half of the points), LHS stratification, and
    use through Geometry.sample_interior(random=...)."""
    import ppsci
    from paddlescience_amd.geometry import sampler

    np.random.seed(5)
    n = 64
    p = sampler.sample(n, ndim, method)
    assert p.shape == (n, ndim) and p.dtype == np.float32
    assert (p > 0).all() and (p < 1).all()
    if method == "Sobol" and ndim >= 3:  # sampler.py:84-88: [0.5, ...] is dropped from three dimensions on
        assert not np.any(np.all(p == 0.5, axis=1))
    for j in range(ndim):
        assert abs(int((p[:, j] < 0.5).sum()) - n // 2) <= 3
    if method == "LHS":
        for j in range(ndim):
            assert sorted(np.floor(p[:, j] * n).astype(int).tolist()) == list(range(n))
    if method == "Halton":
        assert p[0, 0] == 0.5 and p[1, 0] == 0.25 and p[2, 0] == 0.75  # radical inverse base 2 of 1, 2, 3
        if ndim > 1:
            np.testing.assert_allclose(p[:2, 1], [1 / 3, 2 / 3], rtol=1e-6)
    geom = ppsci.geometry.Rectangle((0.0, 0.0), (2.0, 1.0))
    pts = geom.sample_interior(50, random=method)
    assert pts["x"].shape == (50, 1) and (pts["x"] > 0).all() and (pts["x"] < 2).all()
    with pytest.raises(ValueError):
        sampler.sample(4, 2, "Foo")


def test_halton_restatement_is_pinned_to_an_independent_implementation():
    """Third-party pin (the reference's skopt is absent): our radical-inverse Halton points == scipy.stats.qmc.Halton
    (unscrambled) at the indices the reference's call site asks for (sampler.py:71-73, 90-92: skip the all-zero point)."""
    from scipy.stats import qmc

    from paddlescience_amd.geometry import sampler

    for ndim in (1, 2, 3, 5):
        ref = qmc.Halton(d=ndim, scramble=False).random(257 + 1)[1:]
        np.testing.assert_allclose(sampler._halton(257, ndim, 1), ref, rtol=0, atol=1e-15)
        got = sampler.quasirandom(257, ndim, "Halton")
        np.testing.assert_array_equal(got, ref.astype(np.float32))
    np.testing.assert_allclose(sampler.radical_inverse(np.arange(1, 9), 2), [0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875, 0.0625])
    ham = sampler.quasirandom(9, 3, "Hammersley")  # (i / N, phi_2(i), phi_3(i)), i = 1 .. 9, N = 10
    np.testing.assert_allclose(ham[:, 0], np.arange(1, 10) / 10.0, rtol=1e-6)
    np.testing.assert_array_equal(ham[:, 1:], sampler._halton(9, 2, 1).astype(np.float32))


def test_sobol_points_are_the_published_sequence():
    """The first points of the unscrambled 2-D Sobol' sequence with the Joe-Kuo direction numbers (scipy.stats.qmc.Sobol's
    documented example): the reference drops the first one (sampler.py:84-88)."""
    from paddlescience_amd.geometry import sampler

    known = np.array([[0.5, 0.5], [0.75, 0.25], [0.25, 0.75], [0.375, 0.375], [0.875, 0.875], [0.625, 0.125], [0.125, 0.625]])
    np.testing.assert_array_equal(sampler.quasirandom(7, 2, "Sobol"), known.astype(np.float32))
    p3 = sampler.quasirandom(6, 3, "Sobol")  # three dimensions: [0, 0, 0] AND [0.5, 0.5, 0.5] dropped
    assert not np.any(np.all(p3 == 0.5, axis=1)) and not np.any(np.all(p3 == 0.0, axis=1))
    np.testing.assert_array_equal(p3[0, :2], [0.75, 0.25])

