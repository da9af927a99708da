"""CPU-only checks: the C-ABI library loads and exports every symbol declared in
include/tramp_hip.h; host-side logic (model algebra, argument validation, error
conventions) behaves as the reference's; the product path refuses to run without a GPU
instead of falling back to CPU arithmetic."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "tramp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tramp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from tramp_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding knows each of them
    assert set(names) == set(_lib.EXPORTS), set(names) ^ set(_lib.EXPORTS)
    assert "gfx950" in _lib.version()


def test_null_context_is_rejected_not_crashed():
    from tramp_amd import _lib
    lib = _lib.load()
    assert lib.tramp_sync(None) == _lib.EINVAL
    assert lib.tramp_gb_prior_fwd(None, 1, 4, 1, None, None, 0.5, 0.0, 1.0, 0.0, None, None, None, None,
                                  None, None) == _lib.EINVAL
    n = ctypes.c_int(-1)
    assert lib.tramp_device_count(ctypes.byref(n)) == _lib.OK and n.value >= 0


def test_no_cpu_fallback_without_gpu():
    import tramp_amd as T
    if T._lib.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(T._lib.TrampHipError):
        T.GaussBernoulliPrior(size=4, rho=0.5).compute_forward_message(1.0, np.zeros(4))


def test_product_path_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tramp_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f"{f} mentions the oracle"


def test_model_algebra_and_observation():
    import tramp_amd as T
    W = np.random.RandomState(0).randn(6, 12)

    class FakeLinear(T.channels.Channel):      # geometry only: no device needed
        Nx, Nz = 6, 12

        def sample(self, Z):
            return W @ Z

        def second_moment(self, tau_z):
            return 2 * tau_z

    prior = T.GaussBernoulliPrior(size=12, rho=0.3)
    model = (prior @ T.SISOVariable(id="x") @ FakeLinear() @ T.SISOVariable(id="z")
             @ T.GaussianChannel(var=0.1) @ T.SILeafVariable(id="y")).to_model()
    assert model.variable_ids == ["x", "z", "y"]
    assert [f.id for f in model.factors] == ["f_0", "f_1", "f_2"]
    s = model.sample(seed=5)
    assert s["x"].shape == (12,) and s["y"].shape == (6,)
    np.testing.assert_allclose(s["z"], W @ s["x"])
    model.init_second_moments()
    tau = model.get_second_moments()
    assert tau["x"] == pytest.approx(0.3) and tau["z"] == pytest.approx(0.6) and tau["y"] == pytest.approx(0.7)
    with pytest.raises(ValueError):
        T.Model(T.DAG([prior, T.GaussianChannel()]))            # factor-factor edge
    with pytest.raises(ValueError):
        (prior @ T.SISOVariable(id="x") @ T.GaussianChannel() @ T.SISOVariable(id="x")
         @ T.GaussianChannel() @ T.SILeafVariable(id="y")).to_model()     # duplicate ids
    with pytest.raises(NotImplementedError):
        T.models.channel2likelihood(FakeLinear(), None, "y")


def test_scheduler_argument_validation():
    import tramp_amd as T
    prior = T.GaussBernoulliPrior(size=4, rho=0.3)
    model = (prior @ T.SISOVariable(id="x") @ T.GaussianChannel(var=0.1) @ T.SILeafVariable(id="y")).to_model()
    with pytest.raises(ValueError):
        T.ExpectationPropagation(model)      # unobserved leaf: must call to_observed first
    assert T.EarlyStoppingEP().tol == 1e-6 and T.EarlyStopping().tol == 1e-6
    ci = T.CustomInit(a_init=[("x", "bwd", 3.0)], a=1.0)
    assert ci.init_a(None, "x", "bwd") == 3.0 and ci.init_a(None, "x", "fwd") == 1.0


def test_host_closed_forms_match_oracle():
    """One-line scalar closed forms that stay on the host (Gaussian prior / likelihood errors, compute_a_new);
    the Marchenko-Pastur channel is kernel K15 (tests/test_gpu_se.py)."""
    import tramp_amd as T
    from oracle import tramp_oracle as orc
    assert T.GaussianPrior(size=None, var=2.0).compute_forward_error(1.3) == orc.gauss_prior_error(1.3, 2.0)
    assert T.GaussianLikelihood(y=None, var=0.5).compute_backward_error(1.3, 1.0) == orc.gauss_lik_error(1.3, 1.0, 0.5)
    f = T.GaussBernoulliPrior(size=4, rho=0.3)
    assert f.compute_a_new(0.5, 1.0) == orc.a_new(0.5, 1.0)
    assert f.compute_a_new(1e-30, 0.0) == 1e11 and f.compute_a_new(10.0, 5.0) == 1e-11


def test_no_undefined_names_in_entry_points():
    """Static guard for the driver-facing scripts and the package: every name that is read is bound
    somewhere in the module (def / import / assignment / argument / comprehension) or is a builtin.
    (A refactor once dropped a helper bench.py still called; nothing ran it on the CPU side.)"""
    import ast
    import builtins
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    files += sorted(glob.glob(os.path.join(root, "tramp_amd", "*.py"))) + sorted(glob.glob(os.path.join(root, "tools", "*.py")))
    for path in files:
        tree = ast.parse(open(path).read(), path)
        bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
        for node in ast.walk(tree):
            if isinstance(node, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
                bound.add(node.name)
            if isinstance(node, (ast.FunctionDef, ast.Lambda, ast.AsyncFunctionDef)):
                a = node.args
                for arg in a.args + a.kwonlyargs + a.posonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                    bound.add(arg.arg)
            elif isinstance(node, (ast.Import, ast.ImportFrom)):
                for al in node.names:
                    bound.add((al.asname or al.name).split(".")[0])
            elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
                bound.add(node.id)
            elif isinstance(node, ast.ExceptHandler) and node.name:
                bound.add(node.name)
            elif isinstance(node, (ast.Global, ast.Nonlocal)):
                bound.update(node.names)
        missing = sorted({n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)} - bound)
        assert not missing, f"{os.path.relpath(path, root)}: names read but never bound: {missing}"


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/tramp_hip.h is a C header (no C++, no torch types): a C99 program compiled with gcc
    -pedantic links against the shared library and calls it."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "consumer.c"
    src.write_text('#include "tramp_hip.h"\n#include <stdio.h>\n'
                   'int main(void) { int n = -1; int rc = tramp_device_count(&n);\n'
                   '  tramp_ctx* ctx = NULL; int rc2 = tramp_sync(ctx);\n'
                   '  printf("%s|%d|%d|%d\\n", tramp_version(), rc, n >= 0, rc2 == TRAMP_EINVAL); return 0; }\n')
    exe = tmp_path / "consumer"
    libdir = os.path.join(root, "tramp_amd")
    res = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                          str(src), "-o", str(exe), "-L", libdir, "-ltramp_hip", f"-Wl,-rpath,{libdir}"],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    version, rc, n_ok, einval = out.stdout.strip().split("|")
    assert "gfx950" in version and rc == "0" and n_ok == "1" and einval == "1"


def test_tree_model_algebra():
    "dag_algebra.py:91-130: + places graphs side by side, @ splices dangling outputs onto dangling inputs in order"
    import tramp_amd as T
    from tramp_amd.variables import SIMOVariable, MILeafVariable, SILeafVariable, SISOVariable
    prior, gb = T.GaussianPrior(size=8), T.GaussBernoulliPrior(size=8, rho=0.2)
    noise, relu = T.GaussianChannel(var=0.1), T.ReluChannel()
    x, y, z = SIMOVariable(id="x", n_next=2), SILeafVariable(id="y"), MILeafVariable(id="z", n_prev=2)
    m = (prior @ x @ (noise @ y + (relu + gb) @ z)).to_model()
    assert not m.is_chain and len(m.variables) == 3 and len(m.factors) == 4
    assert m.prev_factors(x) == [prior] and set(m.next_factors(x)) == {noise, relu}
    assert set(m.prev_factors(z)) == {relu, gb} and m.next_factors(z) == []
    assert m.prev_factors(y) == [noise]
    order = m.forward_ordering
    assert order.index(prior) < order.index(x) < order.index(relu) < order.index(z)
    assert m.init_shapes() == {"x": (8,), "y": (8,), "z": (8,)}
    chain = (T.GaussBernoulliPrior(size=8) @ SISOVariable(id="x") @ T.ReluChannel() @ SISOVariable(id="z")
             @ T.GaussianChannel(var=1.0) @ SILeafVariable(id="y")).to_model()
    assert chain.is_chain and [v.id for v in chain.variables] == ["x", "z", "y"]
    with pytest.raises(ValueError):                    # dangling output
        (T.GaussianPrior(size=8) @ SIMOVariable(id="x", n_next=2) @ T.ReluChannel()).to_model()
    with pytest.raises(ValueError):                    # two factors in a row
        T.GaussianPrior(size=8) @ T.ReluChannel()


def test_every_environment_switch_is_in_the_one_table():
    """tramp_amd/switches.py lists every TRAMP_* variable the library reads - no more, no fewer - with its default;
    bench.py reports the ones found set (env_overrides), so a result line says by itself that it ran on the defaults."""
    import glob
    import re
    from tramp_amd import switches
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = set()
    pat = re.compile(r'(?:getenv|env_size|environ\.get)\(\s*"(TRAMP_[A-Z0-9_]+)"')
    for path in glob.glob(os.path.join(root, "tramp_amd", "csrc", "*.h*")) + glob.glob(os.path.join(root, "tramp_amd", "*.py")):
        if path.endswith("switches.py"):
            continue
        found |= set(pat.findall(open(path).read()))
    assert found == set(switches.SWITCHES), (sorted(found - set(switches.SWITCHES)), sorted(set(switches.SWITCHES) - found))
    assert switches.overrides({"TRAMP_EW_RESIDENT": "0", "HOME": "/x", "TRAMP_UNKNOWN": "1"}) == {"TRAMP_EW_RESIDENT": "0"}
    assert all(kind in ("select", "tune", "debug") for _, kind, _ in switches.SWITCHES.values())
    assert switches.markdown_table().count("\n") == len(switches.SWITCHES) + 1


def test_bench_compact_line_from_a_full_report():
    """bench.py's stdout line built from round 4's 27 KB report (the one the driver could not parse): below 4 KB,
    strict JSON, carries value / roofline / cpu_baseline and the scalars of the other configs."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(ROOT, "profiles", "r04", "bench_default_N100000_final_library_run4.json")))
    full["mse_after_run"] = float("nan")                       # a NaN in a leg must not produce a non-strict line
    line = bench.compact_line(bench._jsonable(full), "bench_extra.json")
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(text) < bench.LINE_LIMIT
    back = json.loads(text)
    assert back["value"] > 1e8 and 0.5 < back["roofline"]["frac"] < 1 and back["roofline"]["bound"] == "mfma"
    assert back["cpu_baseline"]["value"] > 0 and back["cpu_baseline"]["kind"] == "port" and back["cpu_baseline"]["cores"] == 64
    assert back["config2_cs1bit_ms"] > 0 and back["config4_twolayer_ms"] > 0 and back["int8_digits7_ms"] > 0
    assert back["mse_after_run"] is None and len(back["config"]["workload"]) <= 300
    assert "fused_updates_n1e6.sgn_lik" in back["legs_with_errors"]     # round 4's stale leg is named, not hidden
