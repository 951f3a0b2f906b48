"""ppsci.utils.reader and the file-backed datasets (reference: utils/reader.py:31-260, data/dataset/csv_dataset.py:30-287,
mat_dataset.py, npz_dataset.py): alias / dtype / shape rules, time-stamp filtering and repetition, weight_dict forms, and a
constraint built from an IterableCSVDataset the way examples/cylinder/2d_unsteady does."""
import numpy as np
import pytest

import ppsci
from ppsci.utils import reader


def _csv(path, cols):
    keys = list(cols)
    with open(path, "w") as f:
        f.write(",".join(keys) + "\n")
        for i in range(len(cols[keys[0]])):
            f.write(",".join(repr(float(cols[k][i])) for k in keys) + "\n")


def test_reader_csv_mat_npz_dat(tmp_path):
    import pickle

    import scipy.io as sio

    rng = np.random.default_rng(0)
    cols = {"Points:0": rng.uniform(-1, 1, 7), "Points:1": rng.uniform(-1, 1, 7), "U:0": rng.standard_normal(7)}
    _csv(tmp_path / "a.csv", cols)
    d = reader.load_csv_file(str(tmp_path / "a.csv"), ("x", "u"), {"x": "Points:0", "u": "U:0"})
    assert d["x"].shape == (7, 1) and d["x"].dtype == np.float32
    np.testing.assert_allclose(d["u"][:, 0], cols["U:0"].astype(np.float32))
    with pytest.raises(KeyError):
        reader.load_csv_file(str(tmp_path / "a.csv"), ("nope",))
    sio.savemat(tmp_path / "a.mat", {"t": np.arange(3.0), "k": np.arange(6).reshape(2, 3)})
    m = reader.load_mat_file(str(tmp_path / "a.mat"), ("t", "k"))
    assert m["t"].shape == (3, 1) and m["t"].dtype == np.float32
    assert m["k"].shape == (6, 1) and np.issubdtype(m["k"].dtype, np.integer)  # integer tables keep their dtype
    np.savez(tmp_path / "a.npz", a=np.ones((4, 2)), b=np.arange(4))
    z = reader.load_npz_file(str(tmp_path / "a.npz"), ("a", "bb"), {"bb": "b"})
    assert z["a"].shape == (4, 2) and z["a"].dtype == np.float32 and z["bb"].dtype == np.arange(4).dtype
    with open(tmp_path / "a.dat", "wb") as f:
        pickle.dump({"p": np.ones(3), "q": np.zeros((2, 2))}, f)
    assert set(reader.load_dat_file(str(tmp_path / "a.dat"))) == {"p", "q"}
    with pytest.raises(NotImplementedError):
        reader.load_vtk_file("x", 0.1, (0,), ("x",), ("u",))


def test_csv_dataset_timestamps_and_weights(tmp_path):
    # a table WITHOUT a time column is repeated at every time stamp (time-major, new leading key "t")
    _csv(tmp_path / "bc.csv", {"x": [0.0, 1.0, 2.0], "y": [5.0, 6.0, 7.0], "u": [1.0, 2.0, 3.0]})
    ds = ppsci.data.dataset.IterableCSVDataset(str(tmp_path / "bc.csv"), ("x", "y"), ("u",), weight_dict={"u": 10},
                                               timestamps=(0.5, 1.5))
    inp, lab, w = next(iter(ds))
    assert ds.input_keys == ("t", "x", "y") and ds.num_samples == 6 and len(ds) == 1
    np.testing.assert_array_equal(inp["t"][:, 0], [0.5, 0.5, 0.5, 1.5, 1.5, 1.5])
    np.testing.assert_array_equal(inp["x"][:, 0], [0, 1, 2, 0, 1, 2])
    np.testing.assert_array_equal(lab["u"][:, 0], [1, 2, 3, 1, 2, 3])
    np.testing.assert_array_equal(w["u"], np.full((6, 1), 10, np.float32))
    # a table WITH a time column is filtered to the given stamps, in their order
    _csv(tmp_path / "probe.csv", {"t": [1, 1, 2, 2, 3, 3], "x": [0, 1, 0, 1, 0, 1], "u": [10, 11, 20, 21, 30, 31]})
    ds = ppsci.data.dataset.CSVDataset(str(tmp_path / "probe.csv"), ("t", "x"), ("u",), timestamps=(3.0, 1.0),
                                       weight_dict={"u": lambda d: 1.0 + d["x"]})
    assert len(ds) == 4 and ds.input_keys == ("t", "x")
    np.testing.assert_array_equal(ds.label["u"][:, 0], [30, 31, 10, 11])
    np.testing.assert_array_equal(ds.weight["u"][:, 0], [1, 2, 1, 2])
    item = ds[np.array([0, 3])]
    np.testing.assert_array_equal(item[1]["u"][:, 0], [30, 11])


def test_supervised_constraint_from_a_csv_file(tmp_path):
    _csv(tmp_path / "ic.csv", {"Points:0": [0.0, 0.5, 1.0], "Points:1": [0.1, 0.2, 0.3], "U:0": [1.0, 1.0, 1.0]})
    cst = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "IterableCSVDataset", "file_path": str(tmp_path / "ic.csv"), "input_keys": ("x", "y"),
                     "label_keys": ("u",), "alias_dict": {"x": "Points:0", "y": "Points:1", "u": "U:0"},
                     "weight_dict": {"u": 10}, "timestamps": (1.0,)}},
        ppsci.loss.MSELoss("mean"), name="IC")
    inp, lab, w = next(cst.data_iter)
    assert set(inp) == {"t", "x", "y"} and inp["t"].shape == (3, 1) and float(w["u"][0, 0]) == 10.0
