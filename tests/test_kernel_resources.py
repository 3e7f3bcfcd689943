"""The register budgets of the production kernels, read from the built library's code-object metadata (tools/kernel_resources.py):
the kernels the BASELINE configurations launch must not spill vector registers -- a spill in a row loop is a scratch round trip per
row.  Runs without a GPU: hipcc cross-compiles, the metadata is in the ELF notes."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def kernels():
    import kernel_resources
    from rankfm_amd import _build
    rows = kernel_resources.kernels(_build.build())
    assert rows, "no device code in the library"
    return rows


def test_the_library_carries_gfx950_code_only(kernels):
    assert {r["triple"].split("--")[-1] for r in kernels} == {"gfx950"}, sorted({r["triple"] for r in kernels})


def pick(kernels, prefix):
    out = [r for r in kernels if r["kernel"].startswith(prefix)]
    assert out, prefix
    return out


def test_the_bench_kernel_and_the_warp_kernels_do_not_spill(kernels):
    # config 2 / config 1: BPR with hot-row accumulators, 16-lane row groups, k = 64 / 32 / 16 -- on segment-major item rows (the bench
    # kernel since round 5) and on row-major ones
    for kpl in (1, 2, 4):
        for fresh in ("false", "true"):
            for split in ("true", "false"):
                for r in pick(kernels, "sgd_segments_kernel<16, %d, %s, true, false, %s>" % (kpl, fresh, split)):
                    assert r["vgpr_spill"] == 0 and r["scratch"] == 0 and r["vgpr"] <= 128, r
    # ... and no segment-major BPR instantiation the host can launch (k = 16 ... 96) spills either (k = 128 stays row-major: rfm_api.hip)
    for kpl in (3, 6):
        for r in pick(kernels, "sgd_segments_kernel<16, %d, false, true, false, true>" % kpl):
            assert r["vgpr_spill"] == 0 and r["scratch"] == 0, r
    assert not [r for r in kernels if r["kernel"].startswith("sgd_segments_kernel<16, 8, false, true, false, true>")]
    # configs 3 / 5 and every other factor count: the WARP state machine
    for r in pick(kernels, "sgd_warp_kernel<16, "):
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0, r


def test_the_pipelined_feature_row_loop_fits_three_wavefronts_per_simd(kernels):
    # config 4: 768-thread workgroups = 168 registers, nothing in scratch
    for r in pick(kernels, "sgd_features_fast_kernel<16, 4, "):
        if r["max_wg"] == 768:
            assert r["vgpr"] <= 168 and r["vgpr_spill"] == 0 and r["scratch"] == 0, r
    for r in pick(kernels, "feat_tables_kernel<16, 4, false>"):
        assert r["vgpr_spill"] == 0, r


def test_the_serving_kernels_of_round_6_do_not_spill(kernels):
    # _recommend's product with both operands in registers (padded k = 32 ... 128, with and without the observed-items mask) and the
    # one-wavefront-per-user selection: all of their state is registers by design
    for r in pick(kernels, "scores_blockmax_reg_kernel<"):
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0 and r["lds_static"] == 0, r
    for r in pick(kernels, "select_blocks_wave_kernel<"):
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0, r


def _memory_ops(lib, symbol_part):
    """vector-memory instructions and vmcnt waits of one kernel of the built library, in program order (llvm-objdump on the code object)"""
    import re
    import shutil
    import subprocess
    import tempfile
    import kernel_resources
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    with open(lib, "rb") as f:
        blob = f.read()
    for _, elf in kernel_resources.code_objects(blob):
        if symbol_part.encode() not in elf:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as tmp:
            tmp.write(elf)
            tmp.flush()
            sym = re.search(rb"_ZN3rfm15" + re.escape(symbol_part.encode()) + rb"\w*", elf).group(0).decode()
            text = subprocess.run([objdump, "-d", "--disassemble-symbols=" + sym, tmp.name], capture_output=True, text=True, check=True).stdout
        ops = []
        for line in text.splitlines():
            m = re.search(r"\b(global_load_dword\w*|global_atomic_\w+|s_waitcnt)\b(.*?)(//|$)", line)
            if not m:
                continue
            if m.group(1) == "s_waitcnt":
                c = re.search(r"vmcnt\((\d+)\)", m.group(2))
                if c:
                    ops.append(("wait", int(c.group(1))))
            else:
                ops.append(("atomic" if "atomic" in m.group(1) else "load", m.group(1)))
        return ops
    raise AssertionError("no code object holds " + symbol_part)


@pytest.mark.parametrize("kpl", [1, 2, 3, 4])
@pytest.mark.parametrize("hot", [False, True])
def test_the_deferred_warp_waits_leave_exactly_the_atomics_behind_the_gathers_in_flight(kpl, hot):
    """rfm_sgd_warp.hpp, DEFER: the gathers are issued from inline assembly, a finished row's atomics behind them, and the wait for the
    gathers is `s_waitcnt vmcnt(N)` with N = the atomic INSTRUCTIONS just issued -- written by hand, because the compiler's counter model
    cannot.  A wait that counted one atomic too many would hand registers to the examine phase before their loads have landed, silently.
    This pins the count in the built library: 2 (k/16 + 1) f32 atomics -- and nothing else that vmcnt counts -- between the run of
    gathers and the first counted wait, k/16 + 1 in front of the second, and no compiler-made wait inside the gather run."""
    from rankfm_amd import _build
    nc = 4
    ops = _memory_ops(_build.build(), "sgd_warp_kernelILi16ELi%dELb0ELb%dELb1E" % (kpl, 1 if hot else 0))
    for n in (2 * (kpl + 1), kpl + 1):
        # (program order is not control flow: what sits in FRONT of the n atomics may be another block's; the n themselves are the path's)
        sites = [k for k, op in enumerate(ops) if op == ("wait", n) and k >= n and all(o[0] == "atomic" and "add_f32" in o[1] for o in ops[k - n:k])]
        assert sites, "no `s_waitcnt vmcnt(%d)` behind exactly %d f32 atomics in sgd_warp_kernel<16, %d, false, %s, true>" % (n, n, kpl, hot)
    # the two-row wait: the atomics sit right behind the run of gathers (nc rows of kpl dwords + the bias; + the step scale on a branch)
    k = [k for k, op in enumerate(ops) if op == ("wait", 2 * (kpl + 1)) and all(o[0] == "atomic" for o in ops[k - 2 * (kpl + 1):k])][0]
    run = ops[:k - 2 * (kpl + 1)]
    tail = []
    while run and run[-1][0] == "load":
        tail.append(run.pop())
    assert len(tail) >= nc * kpl + 1, "a wait or an atomic inside the run of gathers: %r" % (ops[max(0, k - 40):k + 1],)
