"""On-disk formats either side of the hot path.

    load_checkpoint(path)         the reference's `@save "./checkpoint/mymodel.bson" p opt l_loss_train ... iter`
                                  (case2/case2.jl:178-184, robertson/rober_crnn.jl:205-207): BSON.jl documents; returns the
                                  numeric entries (p, loss lists, iter) so that training can resume from a reference run.
    save_checkpoint(path, ...)    writes p / loss lists / iter in the same BSON.jl array encoding (the Flux optimiser
                                  object of the reference's files is not reproduced).
    load_exp(filename, beta)      Cathode_NCM333_UQ/src_333/dataset.jl:5-23: CSV [T, replicas...] -> unique rows, time grid
                                  t = (T - 100) * 60 / beta.

Needs the `bson` module (pymongo) for the checkpoint functions; nothing here touches the GPU.
"""
from __future__ import annotations

import numpy as np

_DT = {"Float64": "<f8", "Float32": "<f4", "Int64": "<i8", "Int32": "<i4"}


def _bson():
    try:
        import bson
    except ImportError as e:  # pragma: no cover
        raise ImportError("crnn_amd.io checkpoint functions need the `bson` module (pymongo)") from e
    return bson


def load_checkpoint(path):
    """-> dict name -> float | int | np.ndarray for every numeric entry of a BSON.jl checkpoint (others are skipped)."""
    bson = _bson()
    with open(path, "rb") as f:
        d = bson.decode(f.read())
    refs = d.get("_backrefs", [])

    def res(x):
        while isinstance(x, dict) and x.get("tag") == "backref":
            x = refs[x["ref"] - 1]
        return x

    def scalar(v):
        v = res(v)
        if isinstance(v, dict):           # boxed {tag: "struct", type: Float32/Float64, data: bytes}
            name = res(v["type"])["name"][-1]
            return float(np.frombuffer(v["data"], dtype=_DT[name])[0])
        return v

    def conv(x):
        x = res(x)
        if isinstance(x, (int, float)):
            return x
        if isinstance(x, list):           # Vector{Any}
            return np.array([scalar(v) for v in x], dtype=float)
        if isinstance(x, dict) and x.get("tag") == "array":
            name = res(x["type"])["name"][-1]
            if name in _DT:
                return np.frombuffer(x["data"], dtype=_DT[name]).reshape(x["size"][::-1]).T.copy()
            return np.array([scalar(v) for v in x["data"]], dtype=float)
        raise TypeError

    out = {}
    for k, v in d.items():
        if k == "_backrefs":
            continue
        try:
            out[k] = conv(v)
        except (TypeError, KeyError, ValueError):
            pass                          # non-numeric entry (e.g. the Flux optimiser struct)
    return out


def save_checkpoint(path, **entries):
    """Write numeric entries (float arrays, ints, floats) as a BSON.jl document (`BSON.@load path p iter ...` reads it)."""
    bson = _bson()
    doc = {}
    for k, v in entries.items():
        if isinstance(v, (int, np.integer)):
            doc[k] = int(v)
        elif isinstance(v, (float, np.floating)):
            doc[k] = float(v)
        else:
            a = np.asarray(v, dtype="<f8")
            doc[k] = {"tag": "array", "type": {"tag": "datatype", "params": [], "name": ["Core", "Float64"]},
                      "size": list(a.shape), "data": np.asfortranarray(a).tobytes(order="F")}
    with open(path, "wb") as f:
        f.write(bson.encode(doc))


def load_exp(filename, heating_rate):
    """exp_data [D, 1 + n_replicas] with column 0 converted from temperature [deg C] to time [s]."""
    raw = np.loadtxt(filename, delimiter=",", dtype=np.float64, ndmin=2)
    _, idx = np.unique(raw[:, 0], return_index=True)
    exp_data = raw[np.sort(idx)].copy()          # indexin(unique(T), T): first occurrences, original order
    exp_data[:, 0] = (exp_data[:, 0] - 100.0) * 60.0 / heating_rate
    return exp_data
