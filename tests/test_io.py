"""crnn_amd/io.py: BSON.jl checkpoint reader/writer and the cathode CSV loader (CPU only)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def test_checkpoint_roundtrip(tmp_path):
    pytest.importorskip("bson")
    from crnn_amd.io import load_checkpoint, save_checkpoint
    rng = np.random.default_rng(0)
    p = rng.standard_normal(25)
    M = rng.standard_normal((3, 4))
    path = str(tmp_path / "mymodel.bson")
    save_checkpoint(path, p=p, l_loss_train=np.array([0.5, 0.25]), w=M, iter=17)
    d = load_checkpoint(path)
    assert np.array_equal(d["p"], p) and np.array_equal(d["w"], M) and d["iter"] == 17
    assert np.array_equal(d["l_loss_train"], [0.5, 0.25])


@pytest.mark.needs_reference       # CPU-container only: never part of the GPU box's `-m gpu` session (tests/conftest.py)
@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "case2", "checkpoint", "mymodel.bson")), reason="reference tree not mounted")
def test_reads_the_reference_checkpoints(fx):
    pytest.importorskip("bson")
    from crnn_amd.io import load_checkpoint
    d = load_checkpoint(os.path.join(REF, "case2", "checkpoint", "mymodel.bson"))
    assert np.array_equal(d["p"], np.array(fx["case2_ckpt"]["p"]))          # the vector the golden fixtures pin
    r = load_checkpoint(os.path.join(REF, "robertson", "checkpoint", "mymodel.bson"))
    assert np.array_equal(r["p"], np.array(fx["rober_ckpt"]["p"]))
    assert isinstance(d["iter"], int) and d["iter"] > 0


def test_load_exp_converts_temperature_to_time(tmp_path):
    from crnn_amd.io import load_exp
    path = str(tmp_path / "UNCERT.csv")
    rows = np.array([[100.0, 1.0, 2.0], [110.0, 3.0, 4.0], [110.0, 9.0, 9.0], [130.0, 5.0, 6.0]])
    np.savetxt(path, rows, delimiter=",")
    e = load_exp(path, 5.0)
    assert np.array_equal(e[:, 0], [0.0, 120.0, 360.0])                   # (T - 100) * 60 / beta, duplicates dropped
    assert np.array_equal(e[:, 1:], [[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])


@pytest.mark.needs_reference
@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "Cathode_NCM333_UQ", "exp_data", "UNCERT_cath_1_5.csv")), reason="reference tree not mounted")
def test_committed_cathode_fixture_was_reduced_from_the_reference_csv():
    from crnn_amd.io import load_exp
    with open(os.path.join(HERE, "golden", "fixtures_cathode.json")) as f:
        cfx = json.load(f)
    e = load_exp(os.path.join(REF, "Cathode_NCM333_UQ", "exp_data", "UNCERT_cath_1_5.csv"), 5.0)
    s = cfx["sets"][1]
    assert np.allclose(e[:, 0], s["ts"]) and np.allclose(e[:, 1:].mean(axis=1), s["dbar"])
