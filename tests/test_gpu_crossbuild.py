"""Cross-build bit-identity: the gradient kernels give the same bits whether the library is built
at -O3 (shipped), -O2, -O1 or with the adjoint kernels' phase timers compiled in (-DCRNN_ADJ_PROF: another register
allocation and schedule).  With -ffp-contract=on the floating-point sequence is fixed by the source, so any difference
between the builds is undefined behaviour or a miscompile -- the failure mode round 2 recorded for auto_adj_kernel
("a printf changed its losses", a memory fault after a logically equivalent edit).  Problems: the AutoTsit5 composite on
robertson and case1, Tsit5 adjoint on case1 / case2, Rosenbrock23 adjoint on case2 / robertson, forward tangents on
case2, the primal launch (predictions + losses) on case2, HyChem (adjoint on one lane and on a lane pair, the primal launch, the AutoTsit5
composite), the cathode kernels (adjoint with full / checkpointed tapes, forward tangents, primal, both composites, the chunked
dual-norm gradient); 1 061 trajectories each (ragged last wavefront); per-trajectory losses, return codes, saved counts, accepted /
rejected steps and the index-order batch gradient compared through a digest of their bytes (tools/crossbuild.py).
Integer comparison of bit patterns: no tolerance."""
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(2400)
def test_gradient_kernels_are_bit_identical_across_builds():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import crossbuild
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    results, mismatches = crossbuild.run(json_out=os.path.join(out, "crossbuild.json"))
    assert set(results) == set(crossbuild.VARIANTS)
    for name, r in results.items():
        assert "error" not in r, (name, r)
        assert len(r) == 22 and all(v["n_accept"] > 0 for v in r.values()), (name, r)     # 12 + round 4's HyChem primal / composite and 8 cathode launches
        # the composite really switches on robertson (Tsit5 start, Rosenbrock23 after the detector fires): it takes a small
        # multiple of Rosenbrock23's step count, not Tsit5's ~19 000 per trajectory
        assert r["rober_autotsit5_adjoint"]["n_accept"] < 4 * r["rober_ros23_adjoint"]["n_accept"]
    chk = results["O3chk"]      # the bounds-checked build: checks compiled in, none fired, same bits as the release build
    assert all(v["bounds_checked"] and v["bounds_violations"] == 0 for v in chk.values()), chk
    assert not any(v["bounds_checked"] for v in results["O3"].values())
    assert not mismatches, [(m[0], m[1]) for m in mismatches]
