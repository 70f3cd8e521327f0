"""The C-ABI shared library loads without a GPU and exports exactly what include/ptrwm.h declares; the ctypes
mirrors of its structs have the C layout; argument validation runs before any HIP call.  CPU only."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ptrwm.h")


@pytest.fixture(scope="module")
def engine():
    import ptrwm_hip

    if not os.path.exists(ptrwm_hip.LIB_PATH):
        sys.path.insert(0, ROOT)
        import __graft_entry__

        __graft_entry__.build()
    return ptrwm_hip


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ptrwm_[a-z_0-9]+)\s*\(", src)))


def test_header_functions_are_all_exported_and_bound(engine):
    lib = engine.load_library()
    names = declared_functions()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ptrwm.h but not exported"
    assert sorted(engine.SYMBOLS) == names  # the Python binding covers the whole header, nothing else
    assert lib.ptrwm_abi_version() == engine.ABI_VERSION


def test_struct_layouts_match_the_c_header(engine, tmp_path):
    """Compile a probe with gcc against the real header and compare sizeof/offsetof with the ctypes mirrors
    (both the product binding and the oracle's)."""
    from oracle import oracle as O

    fields = {
        "ptrwm_target_desc": ["kind", "dim", "p", "ip", "vec0", "vec1"],
        "ptrwm_proposal_desc": ["kind", "inv_dim", "temp_scale", "dim_scale"],
        "ptrwm_run_args": [f[0] for f in engine.RunArgs._fields_],
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for s, fs in fields.items():
        lines.append(f'printf("{s} %zu\\n", sizeof({s}));')
        for f in fs:
            lines.append(f'printf("{s}.{f} %zu\\n", offsetof({s}, {f}));')
    lines.append("return 0;}")
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-o", str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for mod in (engine, O):
        for cname, cls in (("ptrwm_target_desc", mod.TargetDesc), ("ptrwm_proposal_desc", mod.ProposalDesc),
                           ("ptrwm_run_args", mod.RunArgs)):
            assert C.sizeof(cls) == int(got[cname])
            for f in fields[cname]:
                assert getattr(cls, f).offset == int(got[f"{cname}.{f}"]), (mod.__name__, cname, f)
    hdr = open(HEADER).read()
    assert int(re.search(r"#define PTRWM_MAX_DIM (\d+)", hdr).group(1)) == engine.MAX_DIM
    assert int(re.search(r"#define PTRWM_MAX_TEMPS (\d+)", hdr).group(1)) == engine.MAX_TEMPS


def test_validation_needs_no_gpu(engine):
    """Every error path returns before the first HIP call, so it can be exercised here."""
    lib = engine.load_library()
    assert engine.strerror(0) == "ok" and "NULL" in engine.strerror(-1)
    assert engine.ext_raw_per_step(engine.PROPOSAL_NORMAL, 30) == 30
    assert engine.ext_raw_per_step(engine.PROPOSAL_UNIFORM_RADIUS, 50) == 51
    with pytest.raises(engine.PTRWMError):
        engine.ext_raw_per_step(7, 3)
    # variants: every dim 1..104, every target and proposal; nothing beyond
    for t in range(10):
        for p in range(3):
            assert all(engine.has_variant(t, p, d) for d in (1, 2, 3, 5, 24, 30, 31, 41, 50, 57, 64, 65, 81, 100, 104))
            assert not engine.has_variant(t, p, 105) and not engine.has_variant(t, p, 0)
    assert not engine.has_variant(10, 0, 30) and not engine.has_variant(0, 3, 30)
    # lane-split variants: every dim, ladders of up to 128 temperatures
    for t in range(10):
        for p in range(3):
            assert all(engine.has_quad_variant(t, p, d, n) for d in (1, 30, 32, 33, 50, 64, 65, 100, 104) for n in (1, 16, 17, 128))
            assert not engine.has_quad_variant(t, p, 30, 129) and not engine.has_quad_variant(t, p, 105, 4)
            # one-thread-per-replica step kernels exist up to dim 64 only
            assert all(engine.has_thread_variant(t, p, d) for d in (1, 30, 50, 64))
            assert not any(engine.has_thread_variant(t, p, d) for d in (0, 65, 80, 100, 104, 105))
    # streaming twins exist exactly where a one-thread-per-replica kernel has the dim compiled in: the common dims for
    # every target, and 9 / 19 / 29 - the reference's HybridRosenbrock data dims - for that target alone (variants.h)
    for t in range(10):
        for p in range(3):
            assert all(engine.has_stream_variant(t, p, d) for d in (2, 3, 4, 5, 10, 20, 30, 50))
            assert not any(engine.has_stream_variant(t, p, d) for d in (1, 6, 8, 21, 31, 48, 64, 65, 100))
            assert all(engine.has_stream_variant(t, p, d) == (t == engine.TARGET_HYBRID_ROSENBROCK) for d in (9, 19, 29))
    # the kernel-form switch: returns the previous setting, rejects unknown values
    assert engine.set_kernel_form(engine.FORM_QUAD) == engine.FORM_AUTO
    assert engine.set_kernel_form(engine.FORM_AUTO) == engine.FORM_QUAD
    with pytest.raises(engine.PTRWMError):
        engine.set_kernel_form(3)

    td, pd, ra = engine.TargetDesc(), engine.ProposalDesc(), engine.RunArgs()
    assert lib.ptrwm_run(None, None, None, None) == -1
    td.kind, td.dim = 0, 30
    ra.struct_size = 4
    assert lib.ptrwm_run(C.byref(td), C.byref(pd), C.byref(ra), None) == -6  # struct size / ABI mismatch
    ra.struct_size = C.sizeof(engine.RunArgs)
    ra.n_temps, ra.swap_every = 257, 1
    assert lib.ptrwm_run(C.byref(td), C.byref(pd), C.byref(ra), None) == -3  # too many temperatures
    ra.n_temps, ra.swap_every = 4, 0
    assert lib.ptrwm_run(C.byref(td), C.byref(pd), C.byref(ra), None) == -5
    ra.swap_every = 1
    assert lib.ptrwm_run(C.byref(td), C.byref(pd), C.byref(ra), None) == 0  # no chains, no steps: a no-op
    ra.n_chains, ra.n_steps = 4, 10
    assert lib.ptrwm_run(C.byref(td), C.byref(pd), C.byref(ra), None) == -1  # state pointers missing
    td.dim = 105
    assert lib.ptrwm_run(C.byref(td), C.byref(pd), C.byref(ra), None) == -2
    td.kind, td.dim = 10, 30
    assert lib.ptrwm_logdensity(C.byref(td), None, None, 1, None) == -4
    td.kind, td.dim = 3, 31  # EvenRosenbrock needs an even dim
    assert lib.ptrwm_logdensity(C.byref(td), None, None, 1, None) == -2
    td.kind, td.dim, td.ip[0], td.ip[1] = 4, 12, 3, 5  # Hybrid: dim must be 1 + n2 (n1 - 1) = 11
    assert lib.ptrwm_logdensity(C.byref(td), None, None, 1, None) == -2
    td.dim = 11
    assert lib.ptrwm_logdensity(C.byref(td), None, None, 0, None) == 0  # empty batch: ok, nothing launched
    # stand-alone swap sweep: same argument block, same error codes
    sa = engine.RunArgs()
    assert lib.ptrwm_swap_sweep(None, 30, 0, 1, None) == -1
    assert lib.ptrwm_swap_sweep(C.byref(sa), 30, 0, 1, None) == -6
    sa.struct_size, sa.n_temps, sa.n_chains = C.sizeof(engine.RunArgs), 8, 4
    assert lib.ptrwm_swap_sweep(C.byref(sa), 105, 0, 1, None) == -2
    assert lib.ptrwm_swap_sweep(C.byref(sa), 30, 0, 0, None) == -5 and lib.ptrwm_swap_sweep(C.byref(sa), 30, 0, 16, None) == -5
    assert lib.ptrwm_swap_sweep(C.byref(sa), 30, -1, 1, None) == -5
    assert lib.ptrwm_swap_sweep(C.byref(sa), 30, 0, 1, None) == -1  # state pointers missing
    sa.n_temps = 1
    assert lib.ptrwm_swap_sweep(C.byref(sa), 30, 0, 1, None) == 0  # one temperature: nothing to exchange
    sa.n_temps = 257
    assert lib.ptrwm_swap_sweep(C.byref(sa), 30, 0, 1, None) == -3
    # split step (user-defined densities): same argument block
    sp, pd2 = engine.RunArgs(), engine.ProposalDesc()
    assert lib.ptrwm_split_propose(None, None, 30, None, None, None) == -1
    assert lib.ptrwm_split_propose(C.byref(pd2), C.byref(sp), 30, None, None, None) == -6
    sp.struct_size, sp.n_temps, sp.n_chains, sp.swap_every = C.sizeof(engine.RunArgs), 8, 4, 1
    assert lib.ptrwm_split_propose(C.byref(pd2), C.byref(sp), 0, None, None, None) == -2
    pd2.kind = 3
    assert lib.ptrwm_split_propose(C.byref(pd2), C.byref(sp), 30, None, None, None) == -4
    pd2.kind = 0
    assert lib.ptrwm_split_propose(C.byref(pd2), C.byref(sp), 30, None, None, None) == -1  # pointers missing
    assert lib.ptrwm_split_accept(C.byref(sp), 30, None, None, None, None) == -1
    sp.swap_every = 0
    assert lib.ptrwm_split_accept(C.byref(sp), 30, None, None, None, None) == -5
    sp.swap_every, sp.n_chains = 1, 0
    assert lib.ptrwm_split_accept(C.byref(sp), 30, None, None, None, None) == 0  # empty batch
    assert lib.ptrwm_split_propose(C.byref(pd2), C.byref(sp), 30, None, None, None) == 0


def test_product_refuses_to_run_without_its_library_or_a_gpu(engine, tmp_path):
    import torch

    with pytest.raises(RuntimeError, match="not found"):
        engine.load_library(str(tmp_path / "nope.so"))
    t = engine.Target(engine.TARGET_ROUGH_CARPET, 3, p=(-1, 0, 1, -1, -1, -1, 0))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.logdensity(t, torch.zeros(2, 3))
    # the run plan validates once, on construction: CPU tensors, wrong shapes and wrong dtypes never reach a launch
    st, lp, b = torch.zeros(2, 4, 3), torch.zeros(2, 4), torch.ones(4)
    prop = engine.Proposal(engine.PROPOSAL_NORMAL, torch.ones(4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.RunPlan(t, prop, state=st, logp=lp, beta=b)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.run(t, prop, state=st, logp=lp, beta=b, step0=0, n_steps=1)
    with pytest.raises(ValueError, match="state dim"):
        engine.RunPlan(t, prop, state=torch.zeros(2, 4, 5), logp=lp, beta=b)
    with pytest.raises(ValueError, match="shapes do not match"):
        engine.RunPlan(t, prop, state=st, logp=torch.zeros(2, 3), beta=b)
    # nothing in the product package imports the oracle
    pkg = os.path.join(ROOT, "rwm-pt-pytorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f


def test_build_gate_holds_for_the_built_objects():
    """tools/kernel_stats.py --check on the objects the library was linked from: no kernel in the register regime hipcc
    miscompiled twice (> 256 VGPRs / any AGPR), no scratch in production step kernels of the max-ILP group.  (The
    Makefile runs the same gate; this keeps it visible in the test report.)"""
    import glob
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    objdir = os.path.join(root, "rwm-pt-pytorch_amd", "build")
    if not glob.glob(os.path.join(objdir, "*.o")):
        pytest.skip("no object files (library built elsewhere)")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("ROCm LLVM tools not installed")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_stats.py"), "--check", "--objdir", objdir],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-500:]
    assert "every kernel <= 256 VGPRs and no AGPRs" in r.stdout


def test_a_compiled_caller_links_and_validates_without_python(engine):
    """tests/capi_host/capi_host_test.cpp: a C++ program against include/ptrwm.h - no Python, no torch in the process.
    Its --symbols mode needs no GPU (ABI version, variant queries, every validation error code)."""
    sys.path.insert(0, ROOT)
    import __graft_entry__

    exe = __graft_entry__.build_capi_host_test()
    out = subprocess.run([exe, "--symbols"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "symbols ok" in out.stdout
    libs = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    names = [ln.split()[0] for ln in libs.splitlines() if ln.strip()]  # sonames (the tree's own path contains "pytorch")
    assert any(n.startswith("libptrwm_hip.so") for n in names)
    assert not any("torch" in n or "python" in n.lower() or n.startswith("libc10") for n in names), names


def test_the_binding_stub_printed_in_integration_md_has_the_c_layout(engine):
    """INTEGRATION.md section B shows the ctypes stub a maintainer of the reference would add.  Documentation drifts:
    execute that very block (library path patched to the built library) and compare its three structures, field by field,
    with the product binding's mirrors, which test_struct_layouts_match_the_c_header pins to the C header."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# algorithms/_ptrwm\.py.*?)```", text, flags=re.S).group(1)
    assert 'C.CDLL("libptrwm_hip.so")' in block
    ns = {}
    exec(block.replace('C.CDLL("libptrwm_hip.so")', f"C.CDLL({engine.LIB_PATH!r})"), ns)
    for name in ("TargetDesc", "ProposalDesc", "RunArgs"):
        doc, ref = ns[name], getattr(engine, name)
        assert C.sizeof(doc) == C.sizeof(ref), name
        assert [f[0] for f in doc._fields_] == [f[0] for f in ref._fields_], name
        for f, *_ in ref._fields_:
            assert getattr(doc, f).offset == getattr(ref, f).offset and getattr(doc, f).size == getattr(ref, f).size, (name, f)
    assert ns["lib"].ptrwm_run.restype is C.c_int32
    # the call shown in the second block uses only fields that exist
    call = re.search(r"```python\n(t = TargetDesc\(kind=0.*?)```", text, flags=re.S).group(1)
    used = set(re.findall(r"[(,\s](\w+)=", re.search(r"a = RunArgs\((.*?)\)\nrc", call, flags=re.S).group(1)))
    assert used <= {f[0] for f in engine.RunArgs._fields_}, used - {f[0] for f in engine.RunArgs._fields_}


def test_auto_form_rule_is_the_fitted_model(engine):
    """The form FORM_AUTO picks (ptrwm_auto_form; csrc/form_table.inc) is the model tools/form_fit.py fits to the committed
    sweep of both forms (profiles/r04_form_sweep_dense.txt): the C++ evaluation and the Python one agree on a grid of
    dims, ladder lengths and batch sizes off the sweep's own points; on the sweep itself the rule always picks the
    faster form; and the saw-tooth is there (the thread form at 1.25 waves per SIMD is slower than at 1.0)."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("form_fit", os.path.join(root, "tools", "form_fit.py"))
    F = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(F)
    tab = F.load(os.path.join(root, "profiles", "r04_form_sweep_dense.txt"))
    model = F.fit(tab)
    assert F.regret(model, tab, "fit data") > 0.99  # near-ties aside, the rule picks the faster form at every point of its own sweep
    simds = 1024  # an MI355X (256 CUs), stated to the library: ptrwm_auto_form_for is a pure function of its arguments
    n = 0
    for dim in (16, 19, 20, 24, 30, 32, 33, 40, 41, 44, 50, 52, 57, 60, 64):
        for T in (1, 2, 5, 8, 16, 20, 32, 48, 64):
            cpw = 64 // T
            for w in (0.1, 0.3, 0.6, 0.9, 1.0, 1.1, 1.4, 1.6, 1.9, 2.0, 2.2, 2.75, 3.5, 4.0, 5.0):
                waves = max(1, int(round(w * simds)))
                C = waves * cpw  # whole waves: w is then exactly waves / simds
                th, qu = F.rates(model, dim, T, waves / simds)
                got = engine.auto_form_for(engine.TARGET_ROUGH_CARPET, engine.PROPOSAL_NORMAL, dim, T, C, simds)
                if abs(th - qu) < 2e-3 * max(th, qu):
                    continue  # a tie within the table's four decimals: either form is right
                assert got == (engine.FORM_QUAD if qu > th else engine.FORM_THREAD), (dim, T, w, C)
                n += 1
    assert n > 2000
    assert engine.auto_form_for(0, 0, 10, 8, 1, simds) == engine.FORM_THREAD      # dim < 16: never lane-split
    assert engine.auto_form_for(0, 0, 100, 8, 10**6, simds) == engine.FORM_QUAD   # dim > 64: the only form
    assert engine.auto_form_for(0, 0, 30, 32, 65536, simds) == engine.FORM_THREAD  # BASELINE configs[2]: 8 waves per SIMD
    assert engine.auto_form_for(0, 0, 30, 32, 65536, 64 * simds) == engine.FORM_QUAD  # the same batch on 64 times the SIMDs: half a wave each
    with pytest.raises(engine.PTRWMError):
        engine.auto_form_for(0, 0, 30, 32, 65536, 0)
    a, q = F.params(model, 30, 1)
    assert a[1] * 1.25 / 2 < a[0]  # the dip of profiles/r02_form_sweep.txt at 81 920 chains


def test_form_table_is_stamped_with_the_sources_it_was_fitted_on(engine):
    """The AUTO form table (csrc/form_table.inc) is a measured artefact: tools/form_fit.py stamps it with the hash of the
    kernel sources the sweep ran on (tools/source_hash.py) and the library carries the hash of the sources it was built
    from.  Equal: the table belongs to this build.  Different: a visible warning - AUTO may pick the slower of two
    bit-identical forms until the sweep is repeated (tools/form_sweep.py dense + tools/form_fit.py); PTRWM_REQUIRE_FRESH_FORM_TABLE=1
    makes it an error (release check)."""
    import importlib.util
    import os
    import warnings

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("source_hash", os.path.join(root, "tools", "source_hash.py"))
    S = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(S)
    built, fitted = engine.source_hash(), engine.form_table_source_hash()
    assert built == S.source_hash(), "the library was not built from the sources in the tree: rebuild (make -C rwm-pt-pytorch_amd/csrc)"
    assert len(built) == 64 and fitted
    if built != fitted:
        msg = (f"csrc/form_table.inc was fitted on other kernel sources ({fitted[:16]}...) than this library is built from "
               f"({built[:16]}...): re-fit the AUTO form rule (tools/form_sweep.py dense > sweep; tools/form_fit.py sweep)")
        assert os.environ.get("PTRWM_REQUIRE_FRESH_FORM_TABLE") != "1", msg
        warnings.warn(msg)

