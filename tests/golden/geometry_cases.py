"""The sampling calls whose results are pinned in tests/golden/geometry.npz.  `run(ns)` executes them
against any implementation exposing Interval / Rectangle / Cuboid / Hypercube / TimeDomain /
TimeXGeometry (the reference under tests/golden/make_geometry_golden.py, ours in tests/test_geometry.py)."""
import numpy as np


def crit_2d(x, y):
    return (x - 0.5) ** 2 + (y - 0.5) ** 2 > 0.04


def crit_txy(t, x, y):
    return x + y < 1.2


def run(ns) -> dict:
    out = {}

    def case(name, seed, fn):
        np.random.seed(seed)
        res = fn()
        if isinstance(res, dict):
            for k, v in res.items():
                out[f"{name}/{k}"] = np.asarray(v)
        else:
            out[name] = np.asarray(res)

    iv = ns.Interval(-1.0, 2.5)
    rect = ns.Rectangle((0.0, 0.0), (1.0, 1.0))
    ldc = ns.Rectangle((-0.05, -0.05), (0.05, 0.05))
    cub = ns.Cuboid((0.0, -1.0, 0.5), (1.0, 1.0, 2.0))
    hc = ns.Hypercube((0.0, 0.0, 0.0, 0.0), (1.0, 2.0, 3.0, 4.0))
    for nm, geo in (("interval", iv), ("rect", rect), ("ldc", ldc), ("cuboid", cub)):
        case(f"{nm}.interior_rand", 1, lambda geo=geo: geo.sample_interior(37))
        case(f"{nm}.interior_even", 2, lambda geo=geo: geo.sample_interior(101, evenly=True))
        case(f"{nm}.interior_sdfd", 3, lambda geo=geo: geo.sample_interior(9, "pseudo", None, False, True))
        case(f"{nm}.boundary_rand", 4, lambda geo=geo: geo.sample_boundary(41))
        case(f"{nm}.boundary_even", 5, lambda geo=geo: geo.sample_boundary(40, evenly=True))
    case("rect.interior_crit", 6, lambda: rect.sample_interior(50, criteria=crit_2d))
    case("rect.boundary_crit", 7, lambda: rect.sample_boundary(30, criteria=lambda x, y: np.isclose(y, 1.0)))
    case("laplace.interior", 8, lambda: rect.sample_interior(10201, evenly=True))
    case("laplace.boundary", 2024, lambda: rect.sample_boundary(400))
    case("ldc.interior_9801", 9, lambda: ldc.sample_interior(9801, evenly=True))
    case("hypercube.random", 10, lambda: hc.random_points(13))
    case("hypercube.boundary", 11, lambda: hc.random_boundary_points(13))
    case("hypercube.uniform", 12, lambda: hc.uniform_points(200))
    case("hypercube.uniform_open", 13, lambda: hc.uniform_points(200, boundary=False))
    tx_step = ns.TimeXGeometry(ns.TimeDomain(0.0, 1.0, time_step=0.25), rect)
    tx_stamps = ns.TimeXGeometry(ns.TimeDomain(0.0, 1.0, timestamps=(0.0, 0.1, 0.5, 1.0)), ns.Interval(-1.0, 1.0))
    tx_plain = ns.TimeXGeometry(ns.TimeDomain(0.0, 2.0), rect)
    case("tx_step.interior", 20, lambda: tx_step.sample_interior(103))
    case("tx_step.interior_crit", 21, lambda: tx_step.sample_interior(64, criteria=crit_txy))
    case("tx_step.interior_even", 22, lambda: tx_step.sample_interior(100, evenly=True))
    case("tx_step.boundary", 23, lambda: tx_step.sample_boundary(50))
    case("tx_step.initial", 24, lambda: tx_step.sample_initial_interior(33))
    case("tx_step.initial_even", 25, lambda: tx_step.sample_initial_interior(36, evenly=True))
    case("tx_stamps.interior", 26, lambda: tx_stamps.sample_interior(31))
    case("tx_stamps.boundary", 27, lambda: tx_stamps.sample_boundary(12))
    case("tx_plain.interior_even", 28, lambda: tx_plain.sample_interior(300, evenly=True))
    case("tx_plain.boundary_even", 29, lambda: tx_plain.sample_boundary(120, evenly=True))
    pts = np.array([[0.0, 0.0], [0.5, 1.0], [1.0, 0.3], [0.2, 0.7], [1.0, 1.0]], dtype="float32")
    out["rect.normal"] = np.asarray(rect.boundary_normal(pts))
    out["rect.on_boundary"] = np.asarray(rect.on_boundary(pts))
    out["rect.sdf"] = np.asarray(rect.sdf_func(pts))
    out["rect.sdf_deriv"] = np.asarray(rect.sdf_derivatives(pts))
    out["cuboid.sdf"] = np.asarray(cub.sdf_func(np.array([[0.5, 0.0, 1.0], [0.0, -1.0, 0.5]], dtype="float32")))
    if hasattr(ns, "Disk"):  # Disk and the boolean (CSG) geometries
        disk = ns.Disk((0.5, 0.5), 0.25)
        big = ns.Rectangle((0.0, 0.0), (2.0, 1.0))
        shapes = {"disk": disk, "diff": big - disk, "union": rect | ns.Disk((1.0, 0.5), 0.4),
                  "inter": big & ns.Disk((0.0, 0.5), 0.7)}
        for i, (nm, geo) in enumerate(shapes.items()):
            case(f"{nm}.interior_rand", 40 + i, lambda geo=geo: geo.sample_interior(53))
            case(f"{nm}.interior_sdfd", 50 + i, lambda geo=geo: geo.sample_interior(7, "pseudo", None, False, True))
            case(f"{nm}.boundary_rand", 60 + i, lambda geo=geo: geo.sample_boundary(47))
        case("disk.boundary_even", 70, lambda: disk.sample_boundary(32, evenly=True))
        case("diff.interior_crit", 71, lambda: shapes["diff"].sample_interior(40, criteria=lambda x, y: x > 0.3))
        cyl = ns.TimeXGeometry(ns.TimeDomain(0.0, 1.0, time_step=0.2), shapes["diff"])
        case("cyl.interior", 72, lambda: cyl.sample_interior(61))
        case("cyl.boundary", 73, lambda: cyl.sample_boundary(29))
        q = np.array([[0.5, 0.75], [0.5, 0.5], [0.0, 0.2], [2.0, 1.0], [0.75, 0.5], [1.3, 0.4]], dtype="float32")
        for nm, geo in shapes.items():
            out[f"{nm}.is_inside"] = np.asarray(geo.is_inside(q))
            out[f"{nm}.on_boundary"] = np.asarray(geo.on_boundary(q))
            out[f"{nm}.sdf"] = np.asarray(geo.sdf_func(q))
        out["diff.normal"] = np.asarray(shapes["diff"].boundary_normal(q[[0, 2, 4]]))
    if hasattr(ns, "Triangle"):  # edge-chain shapes; one triangle given clockwise, one polygon given clockwise
        tri = ns.Triangle((0.0, 0.0), (1.0, 0.0), (0.2, 0.8))
        tri_cw = ns.Triangle((0.1, 0.1), (0.3, 1.2), (1.5, 0.4))
        poly = ns.Polygon(((0, 0), (1, 0), (2, 1), (2, 2), (0, 2)))
        poly_cw = ns.Polygon(((0.0, 0.0), (0.0, 1.5), (0.7, 0.6), (1.6, 1.4), (1.2, -0.3)))
        edge_shapes = {"tri": tri, "tri_cw": tri_cw, "poly": poly, "poly_cw": poly_cw}
        for i, (nm, geo) in enumerate(edge_shapes.items()):
            case(f"{nm}.interior_rand", 90 + i, lambda geo=geo: geo.sample_interior(45))
            case(f"{nm}.boundary_rand", 100 + i, lambda geo=geo: geo.random_boundary_points(33))
            case(f"{nm}.boundary_even", 110 + i, lambda geo=geo: geo.uniform_boundary_points(37))
            case(f"{nm}.random_points", 120 + i, lambda geo=geo: geo.random_points(29))
            np.random.seed(130 + i)
            q = (np.random.random((60, 2)) * 2.6 - 0.4).astype("float32")
            q = np.concatenate([q, np.asarray(geo.uniform_boundary_points(9), dtype="float32")])
            out[f"{nm}.is_inside"] = np.asarray(geo.is_inside(q))
            out[f"{nm}.on_boundary"] = np.asarray(geo.on_boundary(q))
            out[f"{nm}.sdf"] = np.asarray(geo.sdf_func(q))
            out[f"{nm}.meta"] = np.asarray([geo.area, geo.perimeter, geo.diam, *np.ravel(geo.bbox)], dtype="float64")
        case("tri.boundary_sample", 140, lambda: tri.sample_boundary(31))
        mid = np.array([[0.5, 0.0], [0.6, 0.4], [0.1, 0.4], [0.3, 0.3]], dtype="float32")
        out["tri.normal"] = np.asarray(tri.boundary_normal(mid))
        case("tri_diff.interior", 141, lambda: (ns.Rectangle((0.0, 0.0), (1.0, 1.0)) - tri).sample_interior(27))
        case("tri_time.interior", 142, lambda: ns.TimeXGeometry(ns.TimeDomain(0.0, 1.0, time_step=0.5), tri_cw).sample_interior(26))
    if hasattr(ns, "PointCloud"):
        rng = np.random.default_rng(5)
        pts = {"x": rng.uniform(0, 1, (40, 1)).astype("float32"), "y": rng.uniform(-1, 1, (40, 1)).astype("float32"),
               "nu": rng.uniform(0.01, 0.1, (40, 1)).astype("float32")}
        bnd = {k: v[:9].copy() for k, v in pts.items()}
        pc = ns.PointCloud(pts, ("x", "y", "nu"), bnd)
        case("pointcloud.interior_rand", 80, lambda: pc.sample_interior(17))
        case("pointcloud.interior_even", 81, lambda: pc.sample_interior(12, evenly=True))
        case("pointcloud.random_boundary", 82, lambda: pc.random_boundary_points(5))
        probe = np.concatenate([pc.interior[:3], pc.interior[3:5] + 0.01])
        out["pointcloud.is_inside"] = np.asarray(pc.is_inside(probe))
        out["pointcloud.bbox"] = np.stack([np.asarray(b) for b in pc.bbox])
    return out
