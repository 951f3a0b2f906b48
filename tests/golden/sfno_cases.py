"""The configurations of tests/golden/sfno.npz (make_sfno_golden.py writes it, tests/test_sfno.py reads it)."""
CASES = {
    # name: n_modes, hidden, lifting, projection, layers, norm, batch, nlat, nlon   (seeded by the LENGTH of the name)
    # the reference example's shape in small (32 x 64 grid, 32 x 32 modes there): degrees = n_modes[0], orders = n_modes[1] // 2
    "sfno_gn": dict(modes=(8, 8), hidden=6, lift=8, proj=8, layers=2, norm="group_norm", B=2, H=8, W=16),
    # odd latitude count, more orders than a quarter of the longitudes, no norm, three layers, three output channels
    "sfno_plain": dict(modes=(9, 12), hidden=5, lift=7, proj=6, layers=3, norm=None, B=3, H=9, W=14, out=3),
}
