"""On-disk formats either side of the hot path.

    load_checkpoint(path)         the reference's `@save "./checkpoint/mymodel.bson" p opt l_loss_train ... iter`
                                  (case2/case2.jl:178-184, robertson/rober_crnn.jl:205-207): BSON.jl documents; returns the
                                  numeric entries (p, loss lists, iter) so that training can resume from a reference run,
                                  and -- when the file holds the reference's Flux optimiser object -- `opt` (its decoded
                                  fields, load_flux_opt) and `opt_state` (the same state in the library's layout
                                  [m(P) | v(P) | beta1^t, beta2^t | ExpDecay eta, update count], crnn_set_opt_state).
    load_flux_opt(doc)            the `opt` entry of such a document: Flux.Optimise.Optimiser([ExpDecay,] ADAM, WeightDecay)
    save_checkpoint(path, ...)    writes p / loss lists / iter / opt_state in the same BSON.jl array encoding (a Flux
                                  optimiser *object* is not written).
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
            pass                          # non-numeric entry (e.g. the Flux optimiser struct: decoded below)
    if isinstance(d.get("opt"), dict) and "p" in out:
        try:
            out["opt"] = load_flux_opt(d)
            out["opt_state"] = flux_opt_state(out["opt"], np.asarray(out["p"]).size)
        except (TypeError, KeyError, ValueError, IndexError):
            pass                          # some other optimiser object
    return out


def load_flux_opt(doc):
    """Decode the `opt` entry of a reference checkpoint (`@save ... opt`, case2/case2.jl:178, robertson/rober_crnn.jl:201):
    `Flux.Optimiser(ExpDecay(...), ADAMW(...))` = Optimiser([ExpDecay, Optimiser([ADAM, WeightDecay])]) (case2/case2.jl:31-32)
    or `ADAMW(...)` = Optimiser([ADAM, WeightDecay]) (rober_crnn.jl:19).  Flux <= 0.12 keeps the state in IdDicts keyed by
    the parameter array: ADAM -> (m, v, (beta1^t, beta2^t)), ExpDecay -> update count.
    -> dict(adam=dict(eta, beta1, beta2, m, v, beta1_pow, beta2_pow), wd=float,
            expdecay=None | dict(eta, decay, decay_step, clip, count))
    `wd` comes back as the Float64 value of the Float32 literal the reference writes (`1.f-6`)."""
    refs = doc.get("_backrefs", [])

    def res(x):
        while isinstance(x, dict) and x.get("tag") == "backref":
            x = refs[x["ref"] - 1]
        return x

    def num(v):
        v = res(v)
        if isinstance(v, dict) and v.get("tag") == "struct":      # boxed Float32 / Float64
            name = res(v["type"])["name"][-1]
            return float(np.frombuffer(v["data"], dtype=_DT[name])[0])
        if isinstance(v, (int, float)):
            return v
        raise TypeError(type(v))

    def arr(x):
        x = res(x)
        name = res(x["type"])["name"][-1]
        return np.frombuffer(x["data"], dtype=_DT[name]).astype(np.float64)

    def tname(x):
        return res(res(x)["type"])["name"][-1]

    def iddict_first_value(x):          # IdDict{Any,Any}: data = [[keys...], [values...]] with one entry (the vector p)
        return res(res(x)["data"][1][0])

    out = dict(adam=None, wd=0.0, expdecay=None)

    def walk(o):
        o = res(o)
        t = tname(o)
        f = o["data"]
        if t == "Optimiser":
            for member in res(f[0]):
                walk(member)
        elif t == "ExpDecay":           # ExpDecay(eta, decay, step, clip, current::IdDict)
            out["expdecay"] = dict(eta=num(f[0]), decay=num(f[1]), decay_step=int(num(f[2])), clip=num(f[3]),
                                   count=int(num(iddict_first_value(f[4]))))
        elif t == "ADAM":               # ADAM(eta, beta::Tuple, state::IdDict)
            b = res(f[1])["data"]
            st = iddict_first_value(f[2])["data"]
            bp = res(st[2])["data"]
            out["adam"] = dict(eta=num(f[0]), beta1=num(b[0]), beta2=num(b[1]), m=arr(st[0]), v=arr(st[1]),
                               beta1_pow=num(bp[0]), beta2_pow=num(bp[1]))
        elif t == "WeightDecay":
            out["wd"] = num(f[0])
        else:
            raise TypeError(t)

    walk(doc["opt"])
    if out["adam"] is None:
        raise TypeError("no ADAM member")
    return out


def flux_opt_state(opt, n_params):
    """The decoded Flux state in the library's layout (include/crnn_hip.h crnn_opt_state_len):
    [m(P) | v(P) | beta1^t, beta2^t | ExpDecay eta, number of updates]."""
    a, e = opt["adam"], opt["expdecay"]
    if a["m"].size != n_params or a["v"].size != n_params:
        raise ValueError("optimiser state does not belong to p")
    tail = [a["beta1_pow"], a["beta2_pow"], e["eta"] if e else 0.0, float(e["count"]) if e else 0.0]   # as crnn_opt_init leaves them
    return np.concatenate([a["m"], a["v"], np.array(tail)])


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
