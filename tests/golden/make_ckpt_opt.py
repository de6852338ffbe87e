#!/usr/bin/env python3
"""Decode the optimiser objects the reference's own checkpoints hold and commit them as a golden vector
(tests/golden/fixtures_ckpt_opt.json).  Run ONCE in the build container (needs /root/reference and the `bson` module).

    case2/checkpoint/mymodel.bson       `@save ... p opt ...` at case2/case2.jl:178, opt = Flux.Optimiser(ExpDecay(5e-3, 0.5,
                                        500 * n_exp_train, 1e-4), ADAMW(0.005, (0.9, 0.999), 1.f-6)) (case2/case2.jl:31-32),
                                        iter = 3700 epochs x 20 update! calls (case2/case2.jl:194-198) = 74 000 updates
    robertson/checkpoint/mymodel.bson   rober_crnn.jl:201, opt = ADAMW(0.005, (0.9, 0.999), 1.f-6) (rober_crnn.jl:19),
                                        iter = 10 850 epochs x 20 experiments = 217 000 updates

These are the only machine-checkable numbers the reference holds for `update!`: ADAM's running powers
(beta1^t, beta2^t) fix Flux's "state starts at beta, is multiplied after use" convention and the update count; the ExpDecay
fields fix "eta halves when count % step == 0, floored at clip"; WeightDecay's field shows that `1.f-6` stays a Float32 (the
decay the reference applies is 9.999999974752427e-07 * p, not 1e-6 * p).  m and v (which depend on the reference's RNG
stream) are kept too: they let a run resume from the reference's optimiser state (crnn_amd/io.py flux_opt_state).
"""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
spec = importlib.util.spec_from_file_location("crnn_io", os.path.join(HERE, "..", "..", "crnn_amd", "io.py"))
io = importlib.util.module_from_spec(spec)
spec.loader.exec_module(io)          # io.py alone: no GPU library is loaded here

out = {}
for key, path, per_epoch in (("case2", "case2/checkpoint/mymodel.bson", 20), ("robertson", "robertson/checkpoint/mymodel.bson", 20)):
    ck = io.load_checkpoint(os.path.join(REF, path))
    o = ck["opt"]
    a = o["adam"]
    out[key] = dict(
        iter=int(ck["iter"]), updates_per_iter=per_epoch, n_updates=int(ck["iter"]) * per_epoch, n_params=int(ck["p"].size),
        adam=dict(eta=a["eta"], beta1=a["beta1"], beta2=a["beta2"], beta1_pow=a["beta1_pow"], beta2_pow=a["beta2_pow"],
                  beta1_pow_hex=float(a["beta1_pow"]).hex(), beta2_pow_hex=float(a["beta2_pow"]).hex(),
                  m=a["m"].tolist(), v=a["v"].tolist()),
        wd=o["wd"], wd_hex=float(o["wd"]).hex(), expdecay=o["expdecay"], opt_state=ck["opt_state"].tolist())
with open(os.path.join(HERE, "fixtures_ckpt_opt.json"), "w") as f:
    json.dump(out, f)
print({k: (v["n_updates"], v["adam"]["beta1_pow"], v["adam"]["beta2_pow"], v["wd"], v["expdecay"]) for k, v in out.items()})
