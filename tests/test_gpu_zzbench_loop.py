"""The benchmark's own step (bench.TrainLoop: captured chunk hipGraph replayed per chunk + FramePrologue + check_grad + FlatAdamW + in-place
repack, bf16) driven for 30+ optimizer steps on a 4-row slice per chunk of the 512^2 frame pair: the loss and every parameter must stay
finite, no step may be discarded by check_grad, and replaying the captured chunk must be idempotent (round 2's bench went non-finite
because memset nodes inside the captured graph were not ordered against the kernels around them: the first replay was right, every
later one accumulated onto the previous replay's sums -- a 2-step eager test cannot see that)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
CHUNKS = [[0, 128, 256, 384], [64, 192, 320, 448]]  # rows through the frame: object and background in every chunk


def make_loop(use_graph=True, comp=False, spp=128, step_graph=False):
    import bench
    from lab4d_amd import _lib, mlp
    _lib.lib()
    mlp.clear_caches()
    return bench.TrainLoop(torch.device("cuda", 0), 512, spp, CHUNKS, mlp.PREC_BF16, comp=comp, use_graph=use_graph, step_graph=step_graph)


def teardown_function(_):
    from lab4d_amd import mlp
    mlp.FUSED_GRAD_ACCUM = False
    mlp.clear_caches()


def test_replaying_the_captured_chunk_is_idempotent():
    loop = make_loop()
    assert loop.graph is not None
    outs = []
    for rep in range(4):
        loop.opt.zero_grad()
        loop.prologue.zero_grad()
        loop.graph.replay()
        torch.cuda.synchronize()
        leaves = torch.cat([v.grad.reshape(-1) for v in loop.prologue.leaves.values() if v.grad is not None])
        outs.append((loop.st_loss.clone(), loop.opt.flat_grad.clone(), leaves.clone()))
    loss0, g0, l0 = outs[0]
    assert bool(torch.isfinite(loss0).all()) and float(loss0[12]) > 0
    for loss, g, l in outs[1:]:
        assert torch.allclose(loss, loss0, rtol=1e-5, atol=0), (loss, loss0)  # same inputs, same kernels: equal up to the order of the loss sums' atomics
        # gradients are sums of fp32 atomics: equal up to the order of the additions
        assert float((g - g0).abs().max()) <= 1e-3 * float(g0.abs().max())
        assert float((l - l0).abs().max()) <= 1e-3 * float(l0.abs().max())


@pytest.mark.parametrize("use_graph", [True, False])
def test_thirty_two_steps_stay_finite(use_graph):
    loop = make_loop(use_graph=use_graph)
    losses = []
    for _ in range(32):
        last = loop.step()
        losses.append(last[12].clone())
    torch.cuda.synchronize()
    losses = torch.stack(losses).cpu()
    assert bool(torch.isfinite(losses).all()), losses
    assert int(loop.opt.dev_step) == 32, "check_grad discarded %d of 32 steps" % (32 - int(loop.opt.dev_step))
    assert all(bool(torch.isfinite(p).all()) for p in loop.params)
    assert float(losses[-1]) < float(losses[0]), (float(losses[0]), float(losses[-1]))  # and the optimizer does optimise


def test_graph_and_eager_steps_agree():
    """Eight steps replayed from the graph against the same eight steps launched eagerly: the same loss trajectory.  (Parameters are not
    compared element by element: Adam turns the sign of a noise-level gradient into a full +-lr step, so two correct runs that differ in
    the order of their fp32 atomics drift apart by up to steps * lr on such elements.)"""
    def run(use_graph):
        loop = make_loop(use_graph=use_graph)
        out = [loop.step()[12].clone() for _ in range(8)]
        torch.cuda.synchronize()
        return torch.stack(out).cpu()

    la, lb = run(True), run(False)
    assert bool(torch.isfinite(la).all()) and bool(torch.isfinite(lb).all())
    assert float(((la - lb).abs() / lb.abs()).max()) < 0.02, (la, lb)


def test_whole_step_graphs_agree_with_eager_steps_and_stay_finite():
    """Round 5: the step as two hipGraphs (bench.TrainLoop.capture_step: zero + prologue + every chunk + prologue backward | check_grad + AdamW + repack;
    the chunks' random draws batched and eager).  32 replayed steps: finite, none discarded (the device-side step count advances inside graph B), the loss
    falls; the first 8 against 8 eager steps: the same trajectory to the tolerance the per-chunk graph is held to (the draws differ -- batched
    arg-sort instead of per-chunk randperm -- so the comparison is statistical, like the existing one)."""
    loop = make_loop(step_graph=True)
    assert loop.graph_a is not None and loop.graph_b is not None and loop.graph is None
    d0 = int(loop.opt.dev_step)
    # capture_step's eager warm-up must leave no update behind (every rank of a data-parallel run starts its first replay from the SAME weights): the
    # optimizer state is what a loop without any capture starts from
    assert d0 == 0 and loop.opt.steps == 0 and float(loop.opt.m.abs().max()) == 0.0 and float(loop.opt.v.abs().max()) == 0.0
    w_start = loop.opt.flat.clone()
    losses = []
    for _ in range(32):
        losses.append(loop.step()[12].clone())
    torch.cuda.synchronize()
    la = torch.stack(losses).cpu()
    assert bool(torch.isfinite(la).all()), la
    assert int(loop.opt.dev_step) - d0 == 32, "check_grad discarded %d of 32 steps" % (32 - (int(loop.opt.dev_step) - d0))
    assert all(bool(torch.isfinite(p).all()) for p in loop.params)
    assert float(la[-1]) < float(la[0])
    del loop
    eager = make_loop(use_graph=False)
    assert torch.equal(eager.opt.flat, w_start), "the captured loop did not start from the seeded weights"
    lb = torch.stack([eager.step()[12].clone() for _ in range(8)]).cpu()
    assert float(((la[:8] - lb).abs() / lb.abs()).max()) < 0.03, (la[:8], lb)


def test_the_whole_step_graph_is_idempotent_on_its_gradients():
    """Graph A twice on the same random draws: the flat gradient and the prologue's leaf state equal up to the order of fp32 atomics (what a memset
    node ordered wrongly against its neighbours, or a gradient buffer not re-zeroed inside the graph, would break on the second replay)."""
    loop = make_loop(step_graph=True)
    outs = []
    for rep in range(3):
        loop.graph_a.replay()
        torch.cuda.synchronize()
        outs.append((torch.stack(loop.st_losses).clone(), loop.opt.flat_grad.clone()))
    l0, g0 = outs[0]
    assert bool(torch.isfinite(l0).all()) and bool(torch.isfinite(g0).all()) and float(g0.abs().max()) > 0
    for l, g in outs[1:]:
        assert torch.allclose(l, l0, rtol=1e-5, atol=0), (l, l0)
        assert float((g - g0).abs().max()) <= 1e-3 * float(g0.abs().max())


def test_check_grad_discards_a_blown_up_step():
    """Trainer.check_grad (engine/trainer.py:581-604): a step whose pre-clip gradient norm exceeds the threshold leaves weights, moments and
    the step count untouched; so does a non-finite one.  Decided on the device: no host round trip in step()."""
    from lab4d_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(0)
    ps = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in [(7, 5), (33,), (3,)]]
    opt = FlatAdamW(ps, 1e-3)
    ref = [p.detach().clone().requires_grad_(True) for p in ps]
    ropt = torch.optim.AdamW([{"params": [p], "lr": 1e-3} for p in ref], betas=(0.9, 0.999), weight_decay=1e-4)
    scales = [1.0, 1e6, 1.0, float("nan"), 0.5, float("inf"), 2.0]
    for i, sc in enumerate(scales):
        for params, o in ((ps, opt), (ref, ropt)):
            o.zero_grad()
            (sum((p * p).sum() for p in params) * 0.01 * sc).backward()
        before = opt.flat.clone(), opt.m.clone(), opt.v.clone()
        opt.step(max_norm=5.0, skip_above=5.0)
        # the reference: clip, then discard when the pre-clip norm exceeds the threshold (zero_grad -> torch's optimizer skips every parameter)
        tn = torch.nn.utils.clip_grad_norm_(ref, 5.0)
        if bool(tn > 5.0) or not bool(torch.isfinite(tn)):
            ropt.zero_grad()
        ropt.step()
        torch.cuda.synchronize()
        blown = not (sc == sc and abs(sc) < 1e3)
        assert int(opt.skipped) == int(blown), (i, sc, int(opt.skipped), float(opt.norm))
        if blown:
            assert torch.equal(opt.flat, before[0]) and torch.equal(opt.m, before[1]) and torch.equal(opt.v, before[2])
        for a, b in zip(ps, ref):
            assert torch.allclose(a, b, rtol=5e-6, atol=2e-7), (i, sc)
    assert int(opt.dev_step) == 4


def test_psnr_against_the_reference_render_on_w0_and_w1():
    """The second half of BASELINE.json's metric as bench.py reports it (`psnr_vs_ref_db`): rendered colour against the REFERENCE's own render at the bench
    shape, training render on W0 and on W1 (the reference's geometry_init fit) and the eval render on W1, fp32 and bf16.  Measured on MI355X:
    146.6 / 99.3 dB (W0), 146.5 / 97.2 dB (W1 training), 147.9 / 97.8 dB (W1 eval)."""
    import bench
    from lab4d_amd import _lib
    _lib.lib()
    out = bench.psnr_vs_reference(torch.device("cuda", 0))
    for k in ("fp32", "w1_fp32", "eval_w1_fp32"):
        assert out[k] > 135.0, (k, out[k])
    for k in ("bf16", "w1_bf16", "eval_w1_bf16"):
        assert out[k] > 90.0, (k, out[k])


def test_two_ranks_on_one_gpu_keep_their_replicas_identical():
    """The N > 1 code path END TO END with the real kernels on the one GPU of this box (`bench.py --gpus 2 --backend gloo --share-gpu`: the launcher,
    the row partition, each rank's whole-step graphs, the flat-gradient all-reduce over gloo, the timing gather, one JSON line): finite, no step
    discarded, and the two replicas' weights BIT-identical after the steps -- which the round-5 capture warm-up (a local optimizer step per rank with no
    all-reduce) broke.  Not a performance number: the ranks share the GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--res", "128", "--spp", "32", "--chunk-rows", "32",
                          "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and len(d["rank_ms_per_step"]) == 2
    assert d["replicas_identical"] is True
    assert d["params_finite"] and d["steps_discarded_by_check_grad"] == 0 and d["loss_last_chunk"] == d["loss_last_chunk"]
    assert "two hipGraph replays" in d["config"]["launch"]
