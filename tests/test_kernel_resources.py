"""Register / scratch / LDS budget of the kernels whose speed depends on them, read from the shipping library's gfx950 code object (no GPU
needed: the metadata of every kernel the C ABI can launch; an instantiation the library does not hold is compiled on the spot with
-Rpass-analysis=kernel-resource-usage).  These kernels run one wavefront per SIMD with a full
register file; a few more live values turn into scratch memory traffic that nothing hides (+40 ... +90 % kernel time), and
a few more KB of LDS halve the blocks per CU.  Each budget below is what the shipped source compiles to, with the measurement it protects.
"""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "crnn_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc not available")


LLVM = "/opt/rocm/lib/llvm/bin"
_LIB_META = None


def _library_kernels():
    """{demangled name without spaces, up to '(' : resources} of every kernel in the SHIPPING library, read from the gfx950 code object's metadata
    (.hip_fatbin -> clang-offload-bundler -> llvm-readelf --notes): no compile, and it is the binary that runs.  Built on demand like everywhere
    else (crnn_amd/_lib.py: rebuilt when the sources' hash differs)."""
    global _LIB_META
    if _LIB_META is not None:
        return _LIB_META
    import tempfile
    sys.path.insert(0, ROOT)
    from crnn_amd import _lib as L
    so = os.path.join(CSRC, "libcrnn_hip.so")
    assert f"src={L.source_hash()} " in L.lib.crnn_build_info().decode()       # the library next to the sources is theirs
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so, os.devnull])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    ks = []
    for blk in notes.split("  - .agpr_count:")[1:]:
        g = lambda key: re.search(r"\." + key + r":\s+(\S+)", blk).group(1)
        ks.append(dict(mangled=g("name"), agpr=int(blk.split()[0]), total=int(g("vgpr_count")), scratch=int(g("private_segment_fixed_size")),
                       lds=int(g("group_segment_fixed_size"))))
    dem = subprocess.run(["c++filt"], input="\n".join(k["mangled"] for k in ks), capture_output=True, text=True, check=True).stdout.splitlines()
    _LIB_META = {}
    for k, name in zip(ks, dem):
        key = name.replace("void ", "", 1).split("(")[0].replace(" ", "")
        k["vgpr"] = k["total"] - k["agpr"]           # the metadata's .vgpr_count is the unified file's total (architectural + accumulation registers)
        _LIB_META[key] = k
    return _LIB_META


def test_sgpr_literal_constants_are_formed_low_word_first_in_the_shipping_isa(tmp_path):
    """CRNN_SCONST (ros23_kernel.hpp:158-166): a double formed where it is used by two `s_mov_b32` with literals inside an asm statement -- the one
    piece of the kernels the SIMT emulation drops, and (round 5 form) not yet executed by a device.  What can be checked without one: in the
    shipping code object every such constant of fexp_ctl / flog_ctl appears as `s_mov_b32 sN, <low word>` directly followed by
    `s_mov_b32 sM, <high word>` (low word into the first output, high word into the second -- the order `sconst_bits` reassembles them in),
    and never in another order."""
    import struct
    so = os.path.join(CSRC, "libcrnn_hip.so")
    _library_kernels()                                         # (asserts that the library is the sources')
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "k.co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so, os.devnull])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    isa = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    src = open(os.path.join(CSRC, "ros23_kernel.hpp")).read()
    body = src[src.index("double fexp_ctl(double x)"):src.index("double z = __builtin_amdgcn_ldexp(p, (int)dn);")] + src[src.index("double flog_ctl(double x)"):src.index("double fexp_ctl(double x)")]
    consts = [float.fromhex(c) if c.startswith("0x") else float(c) for c in re.findall(r"CRNN_SCONST\(([^)]+)\)", body)]
    assert len(consts) >= 14, consts
    for c in consts:
        b = struct.unpack("<Q", struct.pack("<d", c))[0]
        lo, hi = b & 0xffffffff, b >> 32
        lit = lambda w: (f"0x{w:x}" if w > 64 else str(w))     # (the assembler prints small values as inline constants)
        pairs = re.findall(r"s_mov_b32 s(\d+), %s\b[^\n]*\n\s*s_mov_b32 s(\d+), %s\b" % (lit(lo), lit(hi)), isa)
        assert pairs, (c, hex(lo), hex(hi))
        # (the two outputs are independent 32-bit SGPRs: the compiler usually allocates an aligned pair and otherwise moves them into one.  The
        #  same coefficients also occur in exponentials that do not use CRNN_SCONST, where the compiler forms them itself in any order; so the
        #  statement is existence of the asm's own two-move form, low word first)


def _compile_resources(tmp_path, header, instantiation, flags=()):
    src = tmp_path / "tu.hip"
    src.write_text(f'#include "{header}"\ntemplate __global__ void {instantiation};\n')
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on", "-I", CSRC,
           *flags, "-Rpass-analysis=kernel-resource-usage", "-c", str(src), "-o", str(tmp_path / "tu.o")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    name = instantiation.split("(")[0].split("::")[-1].split("<")[0]
    blocks = re.split(r"remark: Function Name: ", out.stderr)
    for b in blocks:
        if b.startswith("_ZN4crnn") and name in b.split("\n")[0]:
            g = lambda key: int(re.search(key + r": (\d+)", b).group(1))
            return dict(vgpr=g(r"VGPRs"), agpr=g(r"AGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"), lds=g(r"LDS Size \[bytes/block\]"),
                        occ=g(r"Occupancy \[waves/SIMD\]"))
    raise AssertionError("kernel not found in the resource report: " + name)


def _resources(tmp_path, header, instantiation, flags=()):
    """Resources of one kernel instantiation: from the shipping library's code object where the library holds it (every instantiation the C ABI
    can launch: milliseconds), by compiling a one-kernel translation unit with the library's flags otherwise (an experiment's variant; seconds)."""
    key = instantiation.split("(")[0].replace(" ", "")
    lib = _library_kernels()
    if not flags and key in lib:
        return dict(lib[key], source="library")
    return dict(_compile_resources(tmp_path, header, instantiation, flags), source="compiled")


ADJ = "const crnn::SolveParams, const double*, const crnn::AdjParams"
SENS = "const crnn::SolveParams, const double*, const double*"


def test_headline_lane_pair_kernel_has_no_scratch_and_fits_two_blocks_of_lds(tmp_path):
    r = _resources(tmp_path, "ros23_adj2_kernel.hpp", f"crnn::ros23_adj2_kernel<6,3,true,256,1>({ADJ})")
    assert r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 512 and r["lds"] <= 81920, r      # 440 registers, 51 KB (round 5; 404 in profiles/r04g)


def test_one_lane_primal_kernels_have_no_scratch(tmp_path):
    r = _resources(tmp_path, "ros23_adj_kernel.hpp", f"crnn::ros23_adj_kernel<6,3,true,false,256,true,false>({ADJ})")
    assert r["scratch"] == 0, r
    r = _resources(tmp_path, "ros23_adj_kernel.hpp", f"crnn::ros23_adj_kernel<6,3,true,false,256,true,true>({ADJ})")     # finite-difference W
    assert r["scratch"] == 0, r


def test_one_lane_gradient_kernel_keeps_its_small_scratch(tmp_path):
    r = _resources(tmp_path, "ros23_adj_kernel.hpp", f"crnn::ros23_adj_kernel<6,3,true,false,256,false,false>({ADJ})")
    assert r["scratch"] <= 76 and r["lds"] <= 163840, r       # 76 B per lane since round 2, one block per CU


def test_dual_norm_kernel_has_no_scratch_and_two_blocks_per_cu(tmp_path):
    r = _resources(tmp_path, "ros23_sens_kernel.hpp", f"crnn::ros23_sens_kernel<6,3,true,false,3,3,128,26>({SENS})")
    # all of d theta / d p staged (26 rows) and still two blocks of 128 per CU: 2 x 79 984 B <= 160 KB; no scratch at 256 + 230 registers
    assert r["scratch"] == 0 and 2 * r["lds"] <= 163840, r
    # robertson's instantiation (all four chunks in one launch: 44 rows): 60 B of scratch until the controller's exponential formed its
    # constants in SGPRs (fexp_ctl) -- six serialised scratch reloads per attempt
    r = _resources(tmp_path, "ros23_sens_kernel.hpp", f"crnn::ros23_sens_kernel<3,6,false,true,4,3,128,44>({SENS})")
    assert r["scratch"] == 0 and 2 * r["lds"] <= 163840, r


def test_tsit5_dual_norm_kernel_has_no_scratch_and_two_blocks_per_cu(tmp_path):
    """case1's and case2's reference algorithm inside the reference's gradient (tsit5_sens_kernel): 100 144 B of LDS in round 4 (one block of
    two wavefronts per CU, two SIMDs idle), 79 072 / 78 800 B since the rows of d theta / d p, the save times and two of the seven stage
    areas left the LDS: two blocks per CU, a wavefront on every SIMD."""
    for inst in ("6,3,true,false,3,3,128,26", "5,4,false,false,4,3,128,13"):
        r = _resources(tmp_path, "tsit5_sens_kernel.hpp", f"crnn::tsit5_sens_kernel<{inst}>({SENS})")
        assert r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 512 and r["lds"] <= 80000, r


HY = "const crnn::SolveParams, const double*, const crnn::HyParams, const crnn::HySensParams"


def test_cathode_dual_norm_kernel_has_no_scratch(tmp_path):
    """cathode_sens_kernel, the default gradient of config 5 (ForwardDiff's chunks 9 + 8, one lane per trajectory): the first chunk's
    instantiation carried 292 B of scratch per lane at 512 registers in round 4 (~100 scratch instructions per attempt); with the attempt's
    stage tangents parked in LDS until the decision and the controller's constants formed in SGPRs: none, 484 registers, 132 KB of LDS
    per block of 256 (one block per CU, as the registers dictate anyway)."""
    CS = "const crnn::CathodeParams, const crnn::CathSensParams"
    for ch in (0, 1):
        r = _resources(tmp_path, "cathode_sens_kernel.hpp", f"crnn::cathode_sens_kernel<256,{ch}>({CS})")
        assert r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 512 and r["lds"] <= 163840, r


def test_cathode_composite_dual_norm_kernel_has_no_scratch(tmp_path):
    """cathode_sens_auto_kernel (the reference's gradient through AutoTsit5(TRBDF2), nine lanes per trajectory): both chunk instantiations
    without scratch (393 / 395 registers), LDS only for the staged observations."""
    CS = "const crnn::CathodeParams, const crnn::CathSensParams"
    for ch in (0, 1):
        r = _resources(tmp_path, "cathode_sens_auto_kernel.hpp", f"crnn::cathode_sens_auto_kernel<256,{ch}>({CS})")
        assert r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 512 and r["lds"] <= 32768, r



def test_hychem_dual_norm_kernels(tmp_path):
    """The HyChem dual-norm gradient.  hychem_sens2_kernel -- sparse directions, closed-form tangents, one column
    per lane, the primal spread over the group, the trajectory's state in an LDS record -- is the kernel every gradient call of the
    training loop runs: NO scratch (round 4's kernel: 5 236 B; 60 B until the addend constants of its exponentials -- which the compiler
    materialised ahead of the loops and reloaded from scratch at every use -- were formed in SGPRs where they are used: ros23_kernel.hpp
    CRNN_SCONST, fexp_vec_s, fexp_ctl) at 488 of 512 registers, 100 KB of LDS per block of 256 (20 trajectories).  The two-columns-per-lane instantiation costs
    1.45x fewer issue slots per trajectory by the static count (tools/isa_attempt_cost.py) but keeps over a kilobyte of scratch: it
    must keep compiling (the A/B is one template argument away once a device is at hand) and is not what ships.
    hychem_sens_kernel -- dense directions, the fallback for a caller's own directions -- keeps round 4's closed-form / shared-factor
    footprint: two blocks of 128 per CU."""
    fast = _resources(tmp_path, "hychem_sens2_kernel.hpp", f"crnn::hychem_sens2_kernel<9,10,12,256,false>({HY})")
    assert fast["scratch"] == 0 and fast["vgpr"] + fast["agpr"] <= 512 and fast["lds"] <= 163840, fast
    comp = _resources(tmp_path, "hychem_sens2_kernel.hpp", f"crnn::hychem_sens2_kernel<9,10,12,256,true>({HY})")       # through AutoTsit5(Rosenbrock23)
    assert comp["scratch"] == 0 and comp["vgpr"] + comp["agpr"] <= 512 and comp["lds"] <= 163840, comp
    two = _resources(tmp_path, "hychem_sens2_kernel.hpp", f"crnn::hychem_sens2_kernel<9,10,6,256>({HY})")
    assert two["lds"] <= 163840, two
    dense = _resources(tmp_path, "hychem_sens_kernel.hpp", f"crnn::hychem_sens_kernel<9,10,128>({HY})")
    assert dense["scratch"] <= 3700 and 2 * dense["lds"] <= 163840, dense


def test_hychem_finite_difference_primal_kernels(tmp_path):
    """hychem_auto_kernel<..., JFD> (crnn_ctx_set_jacobian on a HyChem context, round 5): ten more inlined point evaluations per stiff
    attempt must not push the lane-pair layout into scratch -- column by column, each evaluation's registers released before the next."""
    for inst in ("true,false", "true,true"):
        r = _resources(tmp_path, "hychem_auto_kernel.hpp", f"crnn::hychem_auto_kernel<9,10,256,{inst}>(const crnn::SolveParams, const double*, const crnn::HyParams)")
        assert r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 512 and r["lds"] <= 160 * 1024, r



def test_headline_kernel_static_issue_budget(tmp_path):
    """The static instruction table of the headline pair kernel by source phase (tools/isa_phase_table.py; profiles/r05d section 1c): round 5 cut the
    reverse sweep's loss + seeds phase from 797 instructions to ~380 executed (766 with both loss-kind specialisations in the table) and the step
    pair from 3 652 to ~3 200 executed.  Budgets with a margin: an edit that brings the branches or the selects back shows up here, without a device."""
    src = tmp_path / "tu.hip"
    src.write_text('#include "ros23_adj2_kernel.hpp"\ntemplate __global__ void crnn::ros23_adj2_kernel<6,3,true,256,1>(' + ADJ + ");\n")
    asm = tmp_path / "k.s"
    out = subprocess.run([HIPCC if os.path.exists(HIPCC) else "hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on", "-I", CSRC,
                          "-gline-tables-only", "-S", "--cuda-device-only", "-o", str(asm), str(src)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    hdr = open(os.path.join(CSRC, "ros23_adj2_kernel.hpp")).read().split("\n")
    args = []
    for k in range(13):
        ln = next(i + 1 for i, l in enumerate(hdr) if f"ADJ2_T({k});" in l)
        args += ["--phase", f"{ln}=p{k}"]
    tab = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_phase_table.py"), str(asm), "ros23_adj2_kernel", "ros23_adj2_kernel.hpp", "--depth", "2", *args],
                         capture_output=True, text=True, timeout=300)
    assert tab.returncode == 0, tab.stderr[-2000:]
    rows = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in tab.stdout.splitlines() if l and l.split()[0] in {f"p{k}" for k in range(13)} | {"total"}}
    seeds, total = rows["p10"], rows["total"]
    assert seeds[0] <= 820 and seeds[1] >= 380, rows          # both specialisations: ~2 x 380 instructions, ~2 x 206 of them FP64 arithmetic
    assert total[0] <= 3750 and total[7] == 0, rows           # the step pair; no scratch instruction inside the loops
    assert rows["p7"][0] <= 290, rows                         # accepted-step bookkeeping (both save-point paths are in the table)
