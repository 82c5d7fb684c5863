"""CPU: the host side's mirror of the weights-stationary dispatch (lab4d_amd.mlp.ws_active / chain_kernel_name) against the rule in
csrc/mlp_kernels.hpp (launch_ws_fwd / launch_ws_bwd) and csrc/mlp_kernels_ws.hpp (ws_ok<Net>(), ws_enabled())."""
import os
import re

from lab4d_amd import mlp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ws_nets_are_the_256_wide_posenc_nets():
    got = set()
    for net in mlp.NET_NAMES:
        d = mlp.describe(net)
        widest = max(max(L.kin, L.mout_pad) for L in d.layers[:d.n_layers])
        if d.emb_kind == 0 and widest == 256 and net not in (mlp.NET_BG_COLOR,):
            got.add(net)
    assert got == set(mlp.WS_NETS), (got, mlp.WS_NETS)


def test_env_switch_follows_atoi(monkeypatch):
    for val, on in ((None, True), ("1", True), ("0", False), ("", False), ("off", False), ("2", True)):
        if val is None:
            monkeypatch.delenv("LAB4D_WS", raising=False)
        else:
            monkeypatch.setenv("LAB4D_WS", val)
        assert mlp.ws_active(mlp.NET_FG_BASE, mlp.PREC_BF16) is on, val
    monkeypatch.delenv("LAB4D_WS", raising=False)
    assert not mlp.ws_active(mlp.NET_FG_BASE, mlp.PREC_F32)
    assert mlp.ws_active(mlp.NET_FG_BASE, mlp.PREC_BF16, dx_only=True)  # the point-gradient-only modes have weights-stationary kernels too
    assert not mlp.ws_active(mlp.NET_VIS, mlp.PREC_BF16)


def test_dispatch_rule_in_the_sources():
    src = open(os.path.join(ROOT, "lab4d_amd", "csrc", "mlp_kernels.hpp")).read()
    # forward: not the point-gradient-only mode; backward: dZ wanted
    assert re.search(r"launch_ws_fwd.*?if \(!ws_enabled\(\)\) return false;", src, re.S)
    assert re.search(r"launch_ws_bwd.*?if \(!ws_enabled\(\)\) return false;", src, re.S)
    assert "k_mlp_fwd_ws<Net, true, false>" in src and "k_mlp_bwd_ws<Net, false>" in src  # point-gradient-only variants
    ws = open(os.path.join(ROOT, "lab4d_amd", "csrc", "mlp_kernels_ws.hpp")).read()
    assert "return e == nullptr || atoi(e) != 0;" in ws


def test_profile_names_of_the_weights_stationary_kernels():
    """rocprofv3 kernel symbols -> the names bench.py reports (tools/pmc_summary.short): the PMC traffic of a kernel is looked up under that name."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(ROOT, "tools", "pmc_summary.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.short("void lab4d::k_mlp_fwd_ws<lab4d::NetFgBase, true, true>(lab4d::FwdK)") == "k_mlp_fwd_ws<FgBase>"
    assert m.short("void lab4d::k_mlp_bwd_ws<lab4d::NetFgColor, true>(lab4d::BwdK)") == "k_mlp_bwd_ws<FgColor>"
    assert m.short("void lab4d::k_mlp_fwd<lab4d::NetFeat, lab4d::PBF16, false, true, true>(lab4d::FwdK)") == "k_mlp_fwd<Feat>"
    assert m.short("void lab4d::k_mlp_wgrad_dma<8, 4, 8, 2>(unsigned short const*)") == "k_mlp_wgrad_dma<8,4>"
    assert mlp.chain_kernel_name("fwd", mlp.NET_FG_BASE, mlp.PREC_BF16) == "k_mlp_fwd_ws<FgBase>"
