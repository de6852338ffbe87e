#!/usr/bin/env python3
"""HyChem smoke run on the golden fixture's three experiments (GPU): prints losses and |grad|."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crnn_amd import NeuralODE, ODEProblem, PRESET_HYCHEM  # noqa: E402

fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_hychem.json")))
a = {k: np.array(fx[k]) for k in ("ts", "u0", "Ttab", "Ptab", "data", "yscale", "dydt_scale", "p")}
node = NeuralODE(ODEProblem(PRESET_HYCHEM, a["ts"], rate_scale=a["dydt_scale"]))
node.set_ensemble(a["u0"], a["data"], a["yscale"])
node.set_tables(a["Ttab"], a["Ptab"])
print("losses", node.losses(a["p"]), "golden", [t["loss"] for t in fx["traj"]], flush=True)
print(node.last_stats, flush=True)
loss, grad = node.loss_and_grad(a["p"])
print("loss", loss, "|grad|", np.linalg.norm(grad), node.last_stats, flush=True)
