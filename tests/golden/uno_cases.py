"""The configurations of tests/golden/uno.npz (make_uno_golden.py writes it, tests/test_uno.py reads it)."""
CASES = {
    # name: dict(hidden, lifting, projection, out_channels, n_modes, scalings, norm, B, H, W, pad, pad_mode, fft_norm, skips_map)
    # (the per-case random stream is seeded by the LENGTH of the name: keep the lengths distinct)
    # the reference config's U (examples/neuraloperator/conf/uno_darcyflow_pretrain.yaml) scaled down: 16 -> pad 19 -> 10 -> 10 -> 20 -> 19
    "uno_cfg": dict(hidden=8, lift=12, proj=10, outs=[6, 8, 8, 8, 6], modes=[[8, 8], [4, 4], [4, 4], [4, 4], [8, 8]],
                    scal=[[1.0, 1.0], [0.5, 0.5], [1, 1], [2, 2], [1, 1]], norm="group_norm", B=2, H=16, W=16, pad=0.2,
                    pad_mode="one-sided", fft_norm="forward", skips=None),
    # even grids without padding: kept columns hit the Nyquist column of one grid and not of the other (16 -> 8 -> 16), rows are
    # cropped away (8 modes on a spectrum cut to 8 rows) -- the irfftn(s=) semantics proper; end-to-end scaling 1
    "uno_even3": dict(hidden=6, lift=8, proj=8, outs=[4, 6, 4], modes=[[16, 16], [8, 8], [8, 8]],
                      scal=[[0.5, 0.5], [1, 1], [2, 2]], norm=None, B=2, H=16, W=16, pad=None, pad_mode="one-sided",
                      fft_norm="backward", skips=None),
    # non-square, anisotropic scaling, ortho norm, an explicit skip map, end-to-end scaling != 1 (12 x 20 -> 18 x 10 -> 18 x 20 -> 9 x 20)
    "uno_aniso_ortho": dict(hidden=5, lift=7, proj=6, outs=[4, 5, 3], modes=[[6, 8], [6, 6], [8, 8]],
                            scal=[[1.5, 0.5], [1.0, 2.0], [0.5, 1.0]], norm="group_norm", B=3, H=12, W=20, pad=None,
                            pad_mode="one-sided", fft_norm="ortho", skips={2: 0}),
}
