"""CPU-side checks on the compiled device code (hipcc cross-compiles gfx950 without a GPU):
 * tools/check_walk26_isa.py -- what the 26-neighbour fast loop relies on beyond the compiler's promises: inline-issued
   loads waited for by a later statement, hard-coded touch registers;
 * every diagnostic -D build the tools use still compiles (-DWA_STAMPS, -DWA_ANT_TIME, -DWA_STRAG_TIME, -DWA_ASM_STAMPS, -DWA_STATE_HASH,
   -DWA_ASM_SPAN_A/B), so the measurement tools cannot rot silently."""
import importlib.util
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

from welding_robot_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _checker():
    spec = importlib.util.spec_from_file_location("check_walk26_isa", os.path.join(ROOT, "tools", "check_walk26_isa.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def device_asm():
    return _checker().assemble()


def test_walk26_loop_keeps_its_loads_and_registers_to_itself(device_asm):
    chk = _checker()
    problems, info = chk.check(device_asm)
    assert not problems, problems
    assert info["touch_loads"] >= 4 and info["deferred_pairs"] >= 1 and info["highest_own_vgpr"] < 200
    lazy = [chk.check_kernel(device_asm.split("\n"), i)[1] for i, l in enumerate(device_asm.split("\n")) if l.startswith("_Z12k_walk_dev26ILb1E")]
    assert lazy and lazy[0]["deferred_singles"] >= 1          # the stamp load of the lazily evaporated field


def test_walk26_checker_sees_what_it_is_there_for(device_asm):
    """the checker on doctored code: the deferred wait removed, a copy of a freshly requested record, a stray use of a touch register"""
    chk = _checker()
    import re
    lines = device_asm.split("\n")
    heads = [i for i, l in enumerate(lines) if re.match(chk.KERNEL_RE, l)]
    assert len(heads) == 2                                   # the dense field's and the lazily evaporated field's instantiation
    for k0 in heads:
        _doctored(chk, lines, k0)


def _doctored(chk, lines, k0):
    k1 = next(i for i in range(k0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[k0:k1]
    w = [i for i, l in enumerate(body) if l.strip() == "s_waitcnt vmcnt(4)"]
    assert w
    no_wait = lines[:k0] + [("\ts_nop 0" if i in w else l) for i, l in enumerate(body)] + lines[k1:]
    assert any("touched before their wait" in p for p in chk.check("\n".join(no_wait))[0])
    t = next(i for i, l in enumerate(body) if "global_load_dword v250" in l)
    stray = lines[:k0] + body[:t] + ["\tv_mov_b32_e32 v3, v251"] + body[t:] + lines[k1:]
    assert any("outside the touch loads" in p for p in chk.check("\n".join(stray))[0])
    # a SINGLE load with a deferred wait (the lazily evaporated field's stamp) is followed like the pairs -- a copy of its register right
    # behind the load, before any wait, must be seen
    _, info = chk.check_kernel(lines, k0)
    for site in info["single_sites"]:
        reg = sorted(chk.vregs(site.split(",")[0]))[0]
        at = [i for i, l in enumerate(body) if l.split(";")[0].strip() == site]
        assert at
        copied = lines[:k0] + body[:at[0] + 1] + ["\tv_mov_b32_e32 v3, v%d" % reg] + body[at[0] + 1:] + lines[k1:]
        assert any("touched before their wait" in p for p in chk.check("\n".join(copied))[0]), site


DIAG_BUILDS = [["-DWA_STAMPS"], ["-DWA_ANT_TIME"], ["-DWA_STRAG_TIME"], ["-DWA_ASM_STAMPS"], ["-DWA_ASM_SPAN_A=2", "-DWA_ASM_SPAN_B=5"],
               ["-DWA_ASM_SPAN_A=10", "-DWA_ASM_SPAN_B=10"], ["-DWA_TEST_KNOBS", "-DWA_STAMPS"], ["-DWA_STATE_HASH", "-DWA_RANK_LDS=64"]]


def test_every_diagnostic_build_compiles(tmp_path):
    def one(defs):
        out = str(tmp_path / ("lib_" + "_".join(d.strip("-D").replace("=", "") for d in defs) + ".so"))
        r = subprocess.run([build.hipcc()] + build.FLAGS + defs + [os.path.join(build.CSRC, "weldacs.hip"), "-o", out, "-L" + build.rocm_lib_dir(), "-lrccl"],
                           capture_output=True, text=True)
        return defs, r.returncode, r.stderr[-1500:]
    with ThreadPoolExecutor(3) as ex:
        for defs, rc, err in ex.map(one, DIAG_BUILDS):
            assert rc == 0, (defs, err)


def test_every_tool_and_example_script_still_compiles():
    """tools/ and examples/ are not exercised by the suites one by one: at least they must parse and name no missing module of the package"""
    import ast
    import glob
    import importlib
    for f in sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "examples", "*.py"))):
        tree = ast.parse(open(f).read(), f)
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("welding_robot_amd"):
                mod = importlib.import_module(node.module)
                for a in node.names:
                    assert hasattr(mod, a.name) or importlib.util.find_spec(node.module + "." + a.name), (f, node.module, a.name)


def test_host_side_writes_to_device_blocks_are_stream_ordered():
    """A block from the context's allocator may still have work queued on the context's (non-blocking) stream -- the 0xff fill of
    WA_DEV_POISON=1, or whatever the previous owner of a recycled block left there.  A synchronous hipMemcpy / hipMemset runs on the null
    stream and does not wait for it (round 5: the "L after i steps" table of acs_create was uploaded that way and, once in ~5 000 solvers
    of the poisoned soak run, lost against the fill).  So: no synchronous host-to-device copy and no synchronous memset in the host code,
    except where a synchronisation stands directly in front (the two counter resets of the debug read-outs)."""
    import glob
    import re
    csrc = os.path.join(ROOT, "welding_robot_amd", "csrc")
    allowed_memset = {"wa_acs_debug_counters", "wa_acs_straggler_counters"}
    bad = []
    for f in sorted(glob.glob(os.path.join(csrc, "*.inc")) + glob.glob(os.path.join(csrc, "*.hip"))):
        fn = None
        for n, line in enumerate(open(f), 1):
            m = re.match(r"^(?:static\s+)?(?:int|int64_t|void|hipError_t)\s+(\w+)\(", line)
            if m:
                fn = m.group(1)
            code = line.split("//")[0]
            if re.search(r"\bhipMemcpy\(", code) and "HostToDevice" in code:
                bad.append("%s:%d %s" % (os.path.basename(f), n, code.strip()))
            if re.search(r"\bhipMemset\(", code) and fn not in allowed_memset:
                bad.append("%s:%d %s" % (os.path.basename(f), n, code.strip()))
    assert not bad, "\n".join(bad)
