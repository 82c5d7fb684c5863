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
    old = {k: os.environ.get(k) for k in ("LAB4D_WS", "LAB4D_CHAIN_GRID")}
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("net", ["fg_base", "fg_color", "dense", "dense6"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_weights_stationary_chains_are_bit_equal_to_the_wave_resident_ones(net, case):
    S, spf, fw, train, dx = CASES[case]
    dx_only = case == 4
    if dx_only and net != "fg_base":
        pytest.skip("point-gradient-only mode: the sdf basefields")
    report = []
    cs = W.make_case(W.NETS[net], S, spf, 11 + case, fw, train, dx, dx_only)
    cs["tangent"] = (net == "fg_base" and case == 0)  # + lab4d_mlp_forward_tangent (the eikonal term's forward) on this case's sign words
    ok = W.compare(cs, "%s case %d" % (net, case), report)
    bad = [b for b in report[-1]["buffers"] if b.get("mismatches") or b.get("ok") is False]
    assert ok, bad


@pytest.mark.parametrize("net", ["fg_base", "fg_color", "dense", "dense6"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_weights_stationary_chains_are_bit_equal_across_tiles(net, case):
    """Round 5: the comparison above launches min(tiles, CUs) workgroups, so every workgroup ran ONE tile -- the persistent loop
    (next-tile weight reload, per-tile bias refill, the last layer's deferred stores flushed into the next tile, the sign-word carry, the posenc scratch
    aliasing an activation buffer, and since round 5 the per-block progress counters that run on across tiles) executed once.  Here the weights-stationary
    launch is forced onto a 3-workgroup grid (LAB4D_CHAIN_GRID, read per launch): 8 .. 38 tiles -> every workgroup runs 2 .. 13 tiles, all five modes
    (training, annealing, tile-uniform frames, inference, point-gradient-only) and the tangent forward, bit for bit against the wave-resident family."""
    S, spf, fw, train, dx = CASES[case]
    dx_only = case == 4
    if dx_only and net != "fg_base":
        pytest.skip("point-gradient-only mode: the sdf basefields")
    os.environ["LAB4D_CHAIN_GRID"] = "3"
    report = []
    cs = W.make_case(W.NETS[net], S, spf, 31 + case, fw, train, dx, dx_only)
    cs["tangent"] = (net == "fg_base" and case == 0)
    ok = W.compare(cs, "%s case %d, 3-workgroup grid" % (net, case), report)
    bad = [b for b in report[-1]["buffers"] if b.get("mismatches") or b.get("ok") is False]
    assert ok, bad


@pytest.mark.parametrize("net", ["fg_base", "fg_color"])
def test_weights_stationary_chains_are_bit_equal_on_a_full_grid_of_several_tiles(net):
    """... and without the override at a size where the default grid (one workgroup per CU, 256) runs 2-3 tiles each: 128 x 600 + 77 samples."""
    os.environ.pop("LAB4D_CHAIN_GRID", None)
    report = []
    cs = W.make_case(W.NETS[net], 128 * 600 + 77, 20000, 41, True, True, True, False)
    ok = W.compare(cs, "%s, 601 tiles" % net, report)
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


def test_a_training_pass_is_the_same_on_both_families():
    """End to end (rays -> warp -> chains -> compositing -> losses -> backward) with LAB4D_WS = 0 / 1: every rendered channel is bit-equal (every chain launch
    is) except the eikonal term, which is formed from d_x; the parameter gradients agree to fp32 summation order (d_x partials per row tile, atomics in the weight-gradient kernels)."""
    from lab4d_amd import deformable as DF
    from lab4d_amd import mlp, synthetic
    M, N, D, res = 2, 64, 16, 64
    P0 = synthetic.make_weights(3)
    g = torch.Generator().manual_seed(5)
    hxy = torch.cat([torch.rand(M, N, 2, generator=g) * res, torch.ones(M, N, 1)], -1).cuda()
    rng = synthetic.to_device({"eik_inds": torch.arange(8), "match_perm": torch.randperm(M * N * D, generator=g)[:1024]}, "cuda")
    res_by = {}
    for ws in ("0", "1"):
        os.environ["LAB4D_WS"] = ws
        mlp.clear_caches()
        Pd = synthetic.to_device(P0, "cuda")
        for v in Pd.values():
            if v.dtype.is_floating_point:
                v.requires_grad_(True)
        frd = synthetic.add_codes(synthetic.to_device(synthetic.make_frames(4, M, res), "cuda"), Pd)
        bd = synthetic.to_device(synthetic.make_targets(6, M, N, res, hxy.cpu()), "cuda")
        frd["feature"] = bd["feature"]
        out = DF.render_train(Pd, frd, hxy, rng, flow_thresh=float(res), n_depth=D, prec=mlp.PREC_BF16)
        loss = sum(DF.losses_fg(out, bd, res, DF.DEFAULT_LOSS_WT).values())
        loss.backward()
        res_by[ws] = ({k: v.detach().clone() for k, v in out["rendered"].items() if torch.is_tensor(v)}, float(loss.detach()),
                      {k: v.grad.detach().clone() for k, v in Pd.items() if v.grad is not None})
    r0, l0, g0 = res_by["0"]
    r1, l1, g1 = res_by["1"]
    for k in r0:
        if k == "eikonal":  # |d sdf / dx| of the drawn rays: a function of d_x, whose row-tile partials add up in another order
            assert float((r0[k] - r1[k]).abs().max()) <= 1e-5 * (float(r0[k].abs().max()) + 1e-20), k
        else:
            assert torch.equal(r0[k], r1[k]), "rendered[%s] differs between the kernel families" % k
    assert abs(l0 - l1) <= 1e-6 * abs(l0), (l0, l1)
    assert set(g0) == set(g1)
    for k in g0:
        den = float(g0[k].abs().max()) + 1e-20
        assert float((g0[k] - g1[k]).abs().max()) <= 2e-3 * den, (k, float((g0[k] - g1[k]).abs().max()), den)
