"""The oracle, the host optimiser and the device optimiser pinned to the one machine-checkable vector the reference
holds for `update!`: the Flux optimiser objects inside its own checkpoints
(case2/checkpoint/mymodel.bson written at case2/case2.jl:178, robertson/checkpoint/mymodel.bson at rober_crnn.jl:201),
decoded once by tests/golden/make_ckpt_opt.py into tests/golden/fixtures_ckpt_opt.json.

What the vector fixes, bit for bit:
  * ADAM's running powers after t updates, (beta1^t, beta2^t) formed by repeated multiplication starting AT beta and
    multiplied AFTER use (Flux <= 0.12): case2 t = 3700 x 20 -> (2.5e-323 [a denormal fixed point of x -> fl(0.9 x)],
    7.009615738047846e-33); robertson t = 10850 x 20 -> (2.5e-323, 5.134646226571125e-95);
  * ExpDecay(5e-3, 0.5, 10000, 1e-4): eta = 1e-4 and the update counter 74 000 ("halve when count % step == 0, floor");
  * WeightDecay holds the Float32 literal `1.f-6`: the factor on p is 9.999999974752427e-07.
m and v depend on the reference's RNG stream and cannot be re-derived; they are carried so that a run can resume from
the reference's optimiser state (crnn_amd.io.flux_opt_state), which the device round trip below exercises.
Integer / bit-pattern comparisons throughout: no tolerance.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


@pytest.fixture(scope="module")
def ck():
    with open(os.path.join(HERE, "golden", "fixtures_ckpt_opt.json")) as f:
        return json.load(f)


def _bits(x):
    return np.asarray(x, np.float64).view(np.uint64)


def _grad_stream(P, seed):
    """Any gradient sequence will do: the pinned tail does not depend on it.  A short cycle keeps the test cheap."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal((16, P)) * 1e-2


CHAINS = {   # reference constructor arguments (case2/case2.jl:31-32, robertson/rober_crnn.jl:19,29)
    "case2": dict(eta=0.005, expdecay=(5e-3, 0.5, 500 * 20, 1e-4)),
    "robertson": dict(eta=0.005, grad_clip_norm=10.0),
}


def _expected_tail(c):
    e = c["expdecay"]
    return np.array([float.fromhex(c["adam"]["beta1_pow_hex"]), float.fromhex(c["adam"]["beta2_pow_hex"]),
                     e["eta"] if e else 0.0, float(e["count"]) if e else 0.0])     # no ExpDecay: the two slots stay 0


def test_fixture_is_self_consistent(ck):
    for key, c in ck.items():
        assert c["n_updates"] == c["iter"] * c["updates_per_iter"]
        assert float.fromhex(c["adam"]["beta1_pow_hex"]) == c["adam"]["beta1_pow"]
        assert float.fromhex(c["adam"]["beta2_pow_hex"]) == c["adam"]["beta2_pow"]
        assert (c["adam"]["eta"], c["adam"]["beta1"], c["adam"]["beta2"]) == (0.005, 0.9, 0.999)
        assert c["wd"] == float(np.float32(1e-6)) != 1e-6          # `1.f-6` (case2.jl:32, rober_crnn.jl:19)
        st = np.array(c["opt_state"])
        P = c["n_params"]
        assert st.size == 2 * P + 4 and np.array_equal(_bits(st[2 * P:]), _bits(_expected_tail(c)))
    assert ck["case2"]["expdecay"] == dict(eta=1e-4, decay=0.5, decay_step=10000, clip=1e-4, count=74000)
    assert ck["robertson"]["expdecay"] is None and ck["case2"]["n_params"] == 25 and ck["robertson"]["n_params"] == 43


@pytest.mark.parametrize("key", ["case2", "robertson"])
def test_oracle_optimiser_reproduces_the_reference_checkpoint_state(orc, ck, key):
    """n_updates calls of the oracle's update! with the reference's constructor arguments end in the reference
    checkpoint's (beta1^t, beta2^t), ExpDecay eta and counter -- bit for bit."""
    c = ck[key]
    P = c["n_params"]
    oopt = orc.Optimiser(P, wd=c["wd"], **CHAINS[key])
    g = _grad_stream(P, 7)
    p = np.full(P, 0.1)
    fn, o, st = orc.lib().orc_opt_update, C.byref(oopt.o), oopt.state
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    pp, sp, gp = dp(p), dp(st), [dp(g[i]) for i in range(g.shape[0])]
    for t in range(c["n_updates"]):
        fn(o, C.c_int(P), pp, gp[t & 15], sp)
    assert np.array_equal(_bits(st[2 * P:]), _bits(_expected_tail(c))), (st[2 * P:], _expected_tail(c))
    assert np.all(np.isfinite(p))


@pytest.mark.parametrize("key", ["case2", "robertson"])
def test_host_optimiser_reproduces_the_reference_checkpoint_state(ck, key):
    """The product's host update (crnn_opt_update, p2vec.hpp opt_update -- the source the device kernel compiles too) with
    the PRESET's constants: the preset must carry the reference's Float32 weight decay, and the chain must end in the
    checkpoint's state."""
    from crnn_amd import Optimiser, PRESET_CASE2, PRESET_ROBER
    from crnn_amd import _lib as L
    c = ck[key]
    P = c["n_params"]
    opt = Optimiser(P, PRESET_CASE2 if key == "case2" else PRESET_ROBER)
    assert _bits(opt.cfg.wd) == _bits(c["wd"]) and opt.cfg.eta == c["adam"]["eta"]
    assert (opt.cfg.beta1, opt.cfg.beta2) == (c["adam"]["beta1"], c["adam"]["beta2"])
    if c["expdecay"]:
        e = c["expdecay"]
        assert (opt.cfg.use_expdecay, opt.cfg.decay_step, opt.cfg.ed_decay, opt.cfg.ed_clip) == (1, e["decay_step"], e["decay"], e["clip"])
    g = _grad_stream(P, 8)
    p = np.full(P, 0.1)
    fn, cfg = L.lib.crnn_opt_update, C.byref(opt.cfg)
    pp, sp, gp = L.dptr(p), L.dptr(opt.state), [L.dptr(g[i]) for i in range(g.shape[0])]
    for t in range(c["n_updates"]):
        fn(cfg, P, pp, gp[t & 15], sp)
    assert np.array_equal(_bits(opt.state[2 * P:]), _bits(_expected_tail(c)))


def test_flux_state_layout(ck):
    """io.flux_opt_state: [m | v | beta powers | ExpDecay eta, count] from decoded Flux fields."""
    from crnn_amd.io import flux_opt_state
    for key, c in ck.items():
        a = c["adam"]
        opt = dict(adam=dict(m=np.array(a["m"]), v=np.array(a["v"]), beta1_pow=a["beta1_pow"], beta2_pow=a["beta2_pow"]),
                   expdecay=c["expdecay"], wd=c["wd"])
        assert np.array_equal(_bits(flux_opt_state(opt, c["n_params"])), _bits(c["opt_state"]))
        with pytest.raises(ValueError):
            flux_opt_state(opt, c["n_params"] + 1)


@pytest.mark.needs_reference
@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present (GPU box)")
def test_fixture_equals_the_reference_files():
    from crnn_amd.io import load_checkpoint
    with open(os.path.join(HERE, "golden", "fixtures_ckpt_opt.json")) as f:
        fxo = json.load(f)
    for key, path in (("case2", "case2/checkpoint/mymodel.bson"), ("robertson", "robertson/checkpoint/mymodel.bson")):
        d = load_checkpoint(os.path.join(REF, path))
        assert np.array_equal(_bits(d["opt_state"]), _bits(fxo[key]["opt_state"]))
        assert d["opt"]["wd"] == fxo[key]["wd"] and d["opt"]["expdecay"] == fxo[key]["expdecay"]
        assert int(d["iter"]) == fxo[key]["iter"]


# ---------------------------------------------------------------------------------------------------------------- GPU
def _ctx(key):
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, PRESET_ROBER, cases
    if key == "case2":
        return NeuralODE(ODEProblem(PRESET_CASE2, cases.case2_tsteps())), PRESET_CASE2
    return NeuralODE(ODEProblem(PRESET_ROBER, cases.rober_tsteps(), rate_scale=np.ones(3))), PRESET_ROBER


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["case2", "robertson"])
def test_device_optimiser_reproduces_the_reference_checkpoint_state(ck, key):
    """The same chain through crnn_train_update (opt_kernel on the device, one launch per update!)."""
    from crnn_amd import Optimiser
    from crnn_amd import _lib as L
    c = ck[key]
    P = c["n_params"]
    node, preset = _ctx(key)
    node.train_init(Optimiser(P, preset), np.full(P, 0.1))
    g = _grad_stream(P, 9)
    gp = [L.dptr(g[i]) for i in range(g.shape[0])]
    fn, h = L.lib.crnn_train_update, node.handle
    for t in range(c["n_updates"]):
        rc = fn(h, gp[t & 15])
        assert rc == 0
    st = node.opt_state()
    assert np.array_equal(_bits(st[2 * P:]), _bits(_expected_tail(c))), (st[2 * P:], _expected_tail(c))
    assert np.all(np.isfinite(node.params()))
    node.close()


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["case2", "robertson"])
def test_resume_from_the_reference_optimiser_state(orc, ck, key):
    """`@load ... opt`: the reference's state (m, v, powers, ExpDecay) goes onto the device as is, and the next update!
    from it equals the oracle's, to the last bit of the state and 1e-15 in p."""
    from crnn_amd import Optimiser
    c = ck[key]
    P = c["n_params"]
    st0 = np.array(c["opt_state"])
    node, preset = _ctx(key)
    p0 = np.linspace(-1.0, 1.0, P)
    node.train_init(Optimiser(P, preset), p0)
    node.set_opt_state(st0)
    assert np.array_equal(_bits(node.opt_state()), _bits(st0))
    oopt = orc.Optimiser(P, wd=c["wd"], **CHAINS[key])
    oopt.state[:] = st0
    g = _grad_stream(P, 10)
    po = p0.copy()
    for i in range(4):
        node.update_(g[i])
        po = oopt.update(po, g[i])
    assert np.max(np.abs(node.params() - po)) < 1e-15
    st = node.opt_state()
    assert np.array_equal(_bits(st[2 * P:]), _bits(oopt.state[2 * P:]))
    assert np.max(np.abs(st[:2 * P] - oopt.state[:2 * P])) <= 1e-15 * np.max(np.abs(oopt.state[:2 * P]))
    node.close()
