"""The weights-stationary chain kernels (csrc/mlp_kernels_ws.hpp) against the wave-resident ones (csrc/mlp_kernels.hpp), launch by launch through
the C ABI (lab4d_mlp_forward / lab4d_mlp_backward with LAB4D_WS = 1 / 0): everything either family writes -- stored embedding, every activation,
every ReLU sign word, the outputs, every dZ, the external gradient -- must be BIT-equal (same packed weights, same accumulation order per output
element, same packed epilogue); d_x to fp32 rounding (its partial sums are formed per row tile and added in a different order).
The wave-resident family is what the oracle-parity tests of test_gpu_mlp.py / test_gpu_field.py pinned in rounds 1-4; with LAB4D_WS at its default
those tests now run the weights-stationary kernels for the 256-wide nets, so both families are held to the oracle and to each other."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location("ws_compare", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ws_compare.py"))
W = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(W)

# (samples, samples per frame, annealing window, training mode, input gradient wanted)
CASES = [(1000, 300, False, True, True),        # ragged tail, tiles that straddle frames (per-lane per-frame bias)
         (128 * 37 + 77, 1000, True, True, True),  # annealing weights, more workgroup tiles than one
         (4096, 2048, False, True, False),      # tiles inside one frame (per-frame bias rows from LDS), no input gradient
         (700, 128, False, False, True),        # inference mode: only the exported layer is stored
         (128 * 21 + 5, 512, True, True, True)]  # point-gradient-only mode (fg_base only): sign words + embedding stored, no activations, no dZ


@pytest.fixture(autouse=True)
def _restore_env():
    old = os.environ.get("LAB4D_WS")
    yield
    if old is None:
        os.environ.pop("LAB4D_WS", None)
    else:
        os.environ["LAB4D_WS"] = old


@pytest.mark.parametrize("net", ["fg_base", "fg_color", "dense", "dense6"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_weights_stationary_chains_are_bit_equal_to_the_wave_resident_ones(net, case):
    S, spf, fw, train, dx = CASES[case]
    dx_only = case == 4
    if dx_only and net != "fg_base":
        pytest.skip("point-gradient-only mode: the sdf basefields")
    report = []
    ok = W.compare(W.make_case(W.NETS[net], S, spf, 11 + case, fw, train, dx, dx_only), "%s case %d" % (net, case), report)
    bad = [b for b in report[-1]["buffers"] if b.get("mismatches") or b.get("ok") is False]
    assert ok, bad


def test_the_dispatch_really_switches_kernels():
    """LAB4D_WS=0 / 1 must launch different kernels (otherwise the comparison above compares a family with itself): timed at a size where the two
    differ measurably, and named differently by the host side."""
    from lab4d_amd import mlp
    os.environ["LAB4D_WS"] = "1"
    assert mlp.chain_kernel_name("fwd", mlp.NET_FG_BASE, mlp.PREC_BF16) == "k_mlp_fwd_ws<FgBase>"
    assert mlp.chain_kernel_name("fwd", mlp.NET_FG_BASE, mlp.PREC_F32) == "k_mlp_fwd<FgBase>"
    assert mlp.chain_kernel_name("bwd", mlp.NET_FEAT, mlp.PREC_BF16) == "k_mlp_bwd<Feat>"
    os.environ["LAB4D_WS"] = "0"
    assert mlp.chain_kernel_name("bwd", mlp.NET_FG_COLOR, mlp.PREC_BF16) == "k_mlp_bwd<FgColor>"
    # inference-mode forward (nothing stored): the weights-stationary kernel is clearly faster there
    c = W.make_case(mlp.NET_FG_BASE, 1 << 20, 1 << 19, 3, False, False, False)
    t = {}
    for ws in (False, True):
        r = W.run_fwd(c, ws)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            W._lib.lib().lab4d_mlp_forward(W.ctypes.byref(r["args"]), W._lib.stream())
        e1.record()
        torch.cuda.synchronize()
        t[ws] = e0.elapsed_time(e1)
    assert t[True] < 0.9 * t[False], t
