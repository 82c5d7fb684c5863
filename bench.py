#!/usr/bin/env python
"""Benchmark of the Lab4D differentiable-volume-rendering hot path on MI355X.

Metric (BASELINE.json): rendered rays/s, forward + backward, at 512x512 x 128 samples/ray, on the
`cat-pikachu` fg-only configuration (Deformable "skel-quad" field: 25 bones, W=256 D=8 SDF/colour MLPs,
visibility / feature / delta-skin MLPs; synthetic weights and frames, there is no data offline).

A *step* is one pass of the hot path over one batch of synthetic input: one 512x512 frame PAIR
(2 x 262,144 rays x 128 samples = 67.1 M samples) through the full training graph
(rays -> backward LBS warp -> visibility / SDF / colour MLPs -> flow + cycle forward warps -> eikonal ->
feature MLP + matching -> gaussian-bone density -> compositing -> losses -> backward to every weight and
per-frame input), gradients all-reduced across ranks (RCCL) and applied with AdamW.  Rays are processed in
chunks with gradient accumulation (per-chunk loss normalisers = the reference's per-rank DDP semantics).
With --gpus N the rows of the frame pair are dealt out to the N ranks round-robin (strong scaling, same total work), and a rank deals
its rows out to its chunks the same way: every chunk then sees the whole image (object and background) like one of the
reference's random pixel batches, instead of a band of empty background at the top of the frame whose masked loss terms have no
positive element (their mean over an empty selection is NaN in the reference, engine/model.py:602, and would poison the step).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver: RCCL needs it (no-op when already exported)
# replayed hipGraphs: memset nodes (PyTorch's multi-block reductions clear their semaphores with one) are only ordered against the kernel
# nodes around them with the runtime's AQL packet capture off; read at HIP initialisation, costs nothing measurable (lab4d_amd/__init__.py)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16 = 2.5e15  # dense MFMA peak, MI355X_MICROARCH.md
PEAK_F32 = 157.3e12
GRAD_SKIP = 5.0 * 1.0  # Trainer.check_grad (engine/trainer.py:581-604): thresh = 5.0 * params_ref["grad_clip"] (1.0 for the parameter group here): a step whose
#                        pre-clip norm exceeds it (or is not finite) is discarded
# algorithmic GEMM FLOPs of the fg training graph per sample, fwd+bwd: 3 x 2 x 918,912 MAC
FLOP_PER_SAMPLE = 5513472.0
# device memory per sample of a training chunk (stored activations, the largest dZ set, masks, per-sample fields): measured 194.9 GiB peak
# for a 16.8 M-sample chunk (128 rows x 2 frames, profiles/r02_bench.json) -- used by --dry-ranks to check that a rank's chunk fits
BYTES_PER_SAMPLE = 194.9 * 2**30 / (2 * 128 * 512 * 128)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks of the data-parallel job, one per GPU.  Without a launcher in front (no WORLD_SIZE in the environment) "
                                                           "and N > 1 this process starts the N ranks itself (torch.distributed.run, like the reference's scripts/train.sh:12-16) "
                                                           "after checking that the box has N GPUs; under a launcher N must equal WORLD_SIZE.  Default: WORLD_SIZE, else 1")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend: nccl (= RCCL on ROCm) for the GPU job; gloo with --dry-step (no GPU) or --share-gpu")
    ap.add_argument("--share-gpu", action="store_true",
                    help="FUNCTIONAL multi-rank run on a box with fewer GPUs than ranks (needs --backend gloo): every rank renders its own rows with the real kernels on "
                         "cuda:(local rank mod device count), the flat gradient is all-reduced over gloo.  Not a performance number (the ranks share a GPU): it exercises the "
                         "N > 1 code path end to end -- partition, per-rank graphs, collective, timing gather -- and checks that the replicas' weights stay identical.")
    ap.add_argument("--dry-step", action="store_true", help="no GPU work: every rank builds its row plan and the product's flat gradient bucket on the CPU, runs the step's ONE "
                                                            "collective (allreduce_flat) through the process group and rank 0 prints the JSON line -- the launch / rendezvous / "
                                                            "collective path of --gpus N exercised where there are no GPUs (CPU tests)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--chunk-rows", type=int, default=None, help="image rows per frame per chunk (default 128 rows x 512 = 65536 rays/frame = 16.8 M samples per chunk, 195 GiB of the 288 GB: "
                                                                "fewer, larger launches -- 64-row chunks measured 2.4 %% slower, 32-row chunks 8 %%)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--config", default="fg", choices=["fg", "comp", "multi", "hash"],
                    help="fg: BASELINE configs[1] (the headline metric).  comp: BASELINE configs[2]'s per-GPU shape -- fg field with the 18-joint human skeleton and "
                         "composed motion (comp_skel-human_dense) + background field, compose_fields, comp losses; spp/2 samples per field.  hash: BASELINE configs[4]'s "
                         "per-GPU shape -- a field on the multiresolution hash encoding (no reference counterpart), 1024x1024, 256 samples/ray (--res / --spp default to those).  "
                         "multi: BASELINE configs[3]'s per-GPU shape -- the fg field of a 10-video category model (num_inst=10: per-instance codes in every CondMLP, "
                         "fg_motion comp_skel-quad_dense: shared quadruped skeleton + dense post-warp), a frame pair of one of the videos")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed training step: no forward-only rate, no PSNR, no fp32 leg (used by the fp32 leg's own child run)")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the short fp32 run of the same step that the default bf16 run appends (fp32_leg)")
    ap.add_argument("--no-graph", action="store_true", help="launch every chunk eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--chunk-graph", action="store_true", help="one captured hipGraph per CHUNK (rounds 1-4) instead of the whole-step pair of graphs "
                                                               "(TrainLoop.capture_step: every chunk + prologue + optimizer in two graph launches per step)")
    ap.add_argument("--hash-f32-table-grad", action="store_true", help="--config hash: every level's table gradient through fp32 atomics (the round-5 form) instead of packed fp16 atomics on the hashed levels")
    ap.add_argument("--hash-no-compact", action="store_true", help="--config hash: evaluate the field on every sample (the round-5 form) instead of the inside-box samples only")
    ap.add_argument("--sustain-steps", type=int, default=75, help="default single-GPU fg run: steps the loop is continued for behind the timed region (the sustained rate, reported beside the headline)")
    ap.add_argument("--cpu-rays", type=int, default=4096, help="rays per frame of one CPU-baseline pass: 2 x 4,096 = one 8,192-ray chunk of the reference's chunking of configs[1]")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group and run the gradient all-reduce even at world size 1 "
                                                              "(exercises init order, graph capture next to RCCL's buffers and the collective call on a 1-GPU box)")
    ap.add_argument("--emulate-rank-of", type=int, default=0,
                    help="single process, no collective: run rank 0's share of an N-rank job (its rows, chunks, prologue, optimizer step) -- the per-rank step "
                         "time an N-GPU run cannot beat; reported under 'emulated', the headline fields stay those of the work actually done")
    ap.add_argument("--trace", action="store_true", help="diagnostic, not a measurement: synchronise after every chunk and optimizer step and print (stderr) the 12 loss "
                                                         "terms of every chunk, the gradient norm, the clip / skip decision and the first parameters that hold a non-finite value")
    ap.add_argument("--poison", action="store_true", help="diagnostic: torch.empty returns NaN-filled memory (torch.utils.deterministic.fill_uninitialized_memory), "
                                                          "so a read of an uninitialised buffer shows up in the first step")
    ap.add_argument("--dry-ranks", type=int, default=0, help="no GPU work: print every rank's row band, chunk list and memory estimate for --gpus N")
    a = ap.parse_args()
    a.res_given, a.spp_given, a.chunk_rows_given = "--res" in sys.argv, "--spp" in sys.argv, a.chunk_rows is not None
    if a.chunk_rows is None:
        a.chunk_rows = 128 if a.dtype == "bf16" else 64  # fp32 activations are twice the size
        if a.config == "multi":
            a.chunk_rows //= 2  # three dense post-warp chains more per sample: a 128-row chunk does not fit 288 GB
    return a


def make_problem(res, device, comp=False, multi=False):
    from lab4d_amd import synthetic
    if multi:  # Deformable("comp_skel-quad_dense", num_inst=10): 25 bones + dense post-warp, 10 instance codes per CondMLP
        w = synthetic.add_dense_weights(synthetic.make_weights(0, num_inst=10), 0, num_inst=10)
        P = synthetic.to_device(w, device)
        for k, v in P.items():
            if v.dtype.is_floating_point and k != "aabb":
                v.requires_grad_(True)
        fr0 = synthetic.make_frames(1, 2, res, num_inst=10)
        fr0["inst_id"] = torch.full((2,), 3, dtype=torch.long)  # both frames of the pair come from video 3
        fr = synthetic.to_device(synthetic.add_codes(fr0, w), device)
        return P, fr
    if comp:  # MultiFields(field_type="comp", fg_motion="comp_skel-human_dense"): 18 bones + dense post-warp
        w = synthetic.add_dense_weights(synthetic.make_weights(0, num_bones=18), 0)
        P = synthetic.to_device(w, device)
        for k, v in P.items():
            if v.dtype.is_floating_point and k != "aabb":
                v.requires_grad_(True)
        fr = synthetic.to_device(synthetic.add_codes(synthetic.make_frames(1, 2, res, num_bones=18), w), device)
        Pb = synthetic.to_device(synthetic.make_bg_weights(0), device)
        for v in Pb.values():
            v.requires_grad_(True)
        frb = synthetic.add_bg_codes(synthetic.to_device(synthetic.make_bg_frames(1, 2, res), device), Pb)
        return P, fr, Pb, frb
    P = synthetic.to_device(synthetic.make_weights(0), device)
    for k, v in P.items():
        if v.dtype.is_floating_point and k != "aabb":
            v.requires_grad_(True)
    fr = synthetic.to_device(synthetic.add_codes(synthetic.make_frames(1, 2, res), synthetic.make_weights(0)), device)
    return P, fr


def chunk_inputs(res, row0, rows, device, seed):
    """Resident inputs of one chunk: rows = a list of image rows, or (with row0) the length of a contiguous band."""
    from lab4d_amd import deformable as DF, synthetic
    hxy = synthetic.make_rays(res, 2, rows=(row0, row0 + rows) if isinstance(rows, int) else list(rows))
    batch = synthetic.to_device(synthetic.make_targets(seed, 2, hxy.shape[1], res, hxy), device)
    # get_mask_balance_wt (engine/model.py:401-424) depends on the targets only: part of the resident input, not of the timed step
    batch["mask_balance_wt"] = DF.mask_balance_wt(batch["mask"], batch["vis2d"], batch["is_detected"])
    batch["mask"] = batch["mask"].float()
    return hxy.to(device), batch


def distinct_indices(S, k, device, gen):
    """k distinct indices, uniform over [0, S), in draw order -- the distribution of randperm(S)[:k] without permuting S elements.  Shapes are
    static (graph-friendly): 2k draws, keep each value's first occurrence, take the first k kept (for S >= 2^20 and k = 1,024 fewer than one
    duplicate is expected among 2k draws)."""
    if S <= 4 * k:
        return torch.randperm(S, device=device, generator=gen)[:k]
    draws = torch.randint(0, S, (2 * k,), device=device, generator=gen)
    vals, order = torch.sort(draws, stable=True)
    first = torch.ones_like(vals, dtype=torch.bool)
    first[1:] = vals[1:] != vals[:-1]
    keep = torch.zeros(2 * k, dtype=torch.bool, device=device)
    keep[order] = first  # draw i survives iff it is the first occurrence of its value
    pos = torch.cumsum(keep.to(torch.int64), 0) - 1
    # (initialised with a VALID index -- the first draw: should fewer than k distinct values survive, the unfilled tail repeats it instead of
    # holding uninitialised int64 gather indices)
    out = draws[:1].expand(2 * k).clone()
    out[torch.where(keep, pos, pos.new_full((), 2 * k - 1))] = draws  # survivors compacted in draw order (the last slot is a dump for the dropped)
    return out[:k]


def draw_rng(M, N, S, device, gen, out=None):
    """Host-independent randomness of one chunk (the reference draws these on the host: nerf.py:438-439, feature.py:177)."""
    eik = torch.randperm(M * N, device=device, generator=gen)[: max(M * N // 16, 1)]
    # the reference draws randperm(S)[:1024] (nnutils/feature.py:177): 1,024 DISTINCT candidates.  A full permutation of 16.8 M indices per chunk is
    # wasted work; 1,024 distinct uniform indices = the first 1,024 distinct values of a uniform stream (over-draw 2x, stable de-duplication)
    perm = distinct_indices(S, min(1024, S), device, gen)
    if out is not None:
        out["eik_inds"].copy_(eik)
        out["match_perm"].copy_(perm)
        return out
    return {"eik_inds": eik, "match_perm": perm}


def draw_rng_all(C, M, N, S, device, gen, outs):
    """draw_rng for the C chunks of a step in one batch of launches (the whole-step graph leaves these draws as the step's only eager work): the same
    distributions -- a uniformly random (M N / 16)-subset in random order per chunk (the prefix of a random permutation = the arg-sort of i.i.d.
    uniforms), 1,024 distinct uniform indices per chunk (distinct_indices, along dim 1) -- into the chunks' static buffers."""
    k_e = max(M * N // 16, 1)
    eik = torch.rand(C, M * N, device=device, generator=gen).argsort(dim=1)[:, :k_e]
    k = min(1024, S)
    if S <= 4 * k:
        perm = torch.rand(C, S, device=device, generator=gen).argsort(dim=1)[:, :k]
    else:
        draws = torch.randint(0, S, (C, 2 * k), device=device, generator=gen)
        vals, order = torch.sort(draws, dim=1, stable=True)
        first = torch.ones_like(vals, dtype=torch.bool)
        first[:, 1:] = vals[:, 1:] != vals[:, :-1]
        keep = torch.zeros_like(first).scatter_(1, order, first)  # draw i survives iff it is the first occurrence of its value
        pos = torch.cumsum(keep.to(torch.int64), 1) - 1
        # (rows initialised with a valid index, their first draw: see distinct_indices)
        out = draws[:, :1].expand(C, 2 * k).clone().scatter_(1, torch.where(keep, pos, pos.new_full((), 2 * k - 1)), draws)  # survivors compacted in draw order
        perm = out[:, :k]
    if isinstance(outs, tuple):  # (stacked (C, k) buffers whose rows the chunks' static dicts are views of: two copies for the whole step)
        outs[0].copy_(eik)
        outs[1].copy_(perm)
        return
    for c, o in enumerate(outs):
        o["eik_inds"].copy_(eik[c])
        o["match_perm"].copy_(perm[c])


def train_chunk(DF, P, fr, hxy, batch, rng, spp, res, prec):
    f = dict(fr)
    f["feature"] = batch["feature"]
    res_d = DF.render_train(P, f, hxy, rng, flow_thresh=float(res), n_depth=spp, prec=prec)
    losses = DF.losses_fg(res_d, batch, res, DF.DEFAULT_LOSS_WT)
    losses.total.backward()  # the sum of the weighted terms, formed by the loss kernel
    return losses.vec.detach()  # the 12 weighted terms + their total


# algorithmic GEMM FLOPs of the comp training graph per RAY at D samples per field, fwd+bwd: fg 918,912 MAC/sample + three
# dense post-warps of 117,248 MAC (backward map once, forward map twice), bg 162,432 MAC/sample
def comp_flop_per_ray(d_field):
    return 3 * 2 * d_field * (918912 + 3 * 117248 + 162432)


def multi_flop_per_ray(d):
    return 3 * 2 * d * (918912 + 3 * 117248)  # the fg field with the dense post-warp in all three warps


def train_chunk_comp(DF, P, fr, Pb, frb, hxy, batch, rng, spp, res, prec):
    f = dict(fr)
    f["feature"] = batch["feature"]
    r = dict(rng, eik_inds_bg=rng["eik_inds"])
    out = DF.render_train_comp(P, f, Pb, frb, hxy, r, flow_thresh=float(res), n_depth=spp // 2, prec=prec)
    losses = DF.losses_comp(out, batch, res, DF.DEFAULT_LOSS_WT)
    losses.total.backward()  # formed by the loss kernel
    return losses.vec.detach()


def eval_rate(DF, P, fr, inputs, spp, prec, use_graph):
    """Forward-only rate of the renderer: render_eval = importance sampling (spp/2 coarse samples -> density -> inverse-CDF
    -> spp samples), backward warp, visibility, normals / eikonal on every sample (one first-order backward of the sdf chain), colour /
    density on the VALID samples only (device-side mask + stream compaction + scatter, no host sync), compositing.  The whole call is
    captured once as a hipGraph and replayed.  FLOPs = the GEMM work actually executed (2 FLOP/MAC), MFMA-bound: fraction of the
    dense bf16 peak."""
    half = inputs[0][0].shape[1] // 2
    n_ev = min(4, len(inputs))
    pick = [inputs[(2 * i + 1) * len(inputs) // (2 * n_ev)] for i in range(n_ev)]  # each chunk's rows are spread over the whole frame
    ev_in = [h[:, :half].contiguous() for h, _ in pick]
    st = ev_in[0].clone()
    counts = []
    for h in ev_in:
        counts.append(DF.render_eval(P, fr, h, n_depth=spp, prec=prec)["debug"]["valid_count"])
    torch.cuda.synchronize()
    valid_frac = sum(float(c) for c in counts) / (n_ev * st.shape[0] * st.shape[1] * spp)
    graph, launch = None, "eager"
    if use_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                DF.render_eval(P, fr, st, n_depth=spp, prec=prec)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                DF.render_eval(P, fr, st, n_depth=spp, prec=prec)
            launch = "hipGraph replay per call"
        except Exception as e:
            graph, launch = None, "eager (capture failed: %s)" % repr(e)[:120]
            torch.cuda.synchronize()
    reps = 3

    def run():
        for _ in range(reps):
            for h in ev_in:
                if graph is not None:
                    st.copy_(h)
                    graph.replay()
                else:
                    DF.render_eval(P, fr, h, n_depth=spp, prec=prec)

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rays = reps * n_ev * st.shape[0] * st.shape[1]
    # per-kernel table of the same calls, launched eagerly (a replayed graph cannot host events): HIP events around every library entry point
    from lab4d_amd import _lib
    _lib.PROF = {}
    for h in ev_in:
        DF.render_eval(P, fr, h, n_depth=spp, prec=prec)
    torch.cuda.synchronize()
    prof = _lib.prof_summary()
    _lib.PROF = None
    kern = {k: round(v[1] / n_ev, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
    skin, base, color, vis = 20736, 573184, 158464 + 37248, 10240
    mac_per_ray = (spp // 2) * (skin + base) + spp * (skin + vis + 2 * base + skin) + valid_frac * spp * (base + color)
    tflops = rays / dt * mac_per_ray * 2 / 1e12
    peak = PEAK_BF16 if prec == 1 else PEAK_F32
    # roofline of the path's dominant kernel (inference-mode chain kernels store nothing: MFMA-bound by construction)
    roof = None
    ranked = sorted(((k, v) for k, v in prof.items() if v[2] > 0), key=lambda kv: -kv[1][1])
    if ranked:
        name, (launches, ms, flops, nbytes) = ranked[0]
        roof = {"bound": "mfma", "kernel": name + " (inference mode)", "achieved": round(flops / (ms * 1e-3) / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                "frac": round(flops / (ms * 1e-3) / peak, 4), "launches": launches, "avg_ms": round(ms / launches, 4), "traffic": None,
                "measured": "HIP events around every launch in an eager re-run of the %d calls" % n_ev}
    return {"value": round(rays / dt, 1), "unit": "rays/s", "launch": launch, "valid_fraction": round(valid_frac, 4),
            "tflops": round(tflops, 1), "frac_of_mfma_peak": round(tflops * 1e12 / peak, 4),
            "kernels_ms_per_call": kern, "ms_per_call": round(dt / (reps * n_ev) * 1e3, 3), "roofline": roof,
            "what": "render_eval: importance sampling (%d coarse + %d fine samples), backward warp, visibility, normals on every sample, colour / "
                    "density on the valid samples (device-side compaction), compositing; %d calls of %d rays" % (spp // 2, spp // 2, reps * n_ev, st.shape[0] * st.shape[1])}


def psnr_vs_reference(dev):
    """Second half of BASELINE.json's metric: PSNR of the rendered colour against the REFERENCE's own render on identical rays / weights.  The
    reference cannot run on the GPU box, so this uses the committed bench-shape fixtures (tests/golden/make_golden.py): a 2-row band of a 512x512
    frame pair x 128 samples/ray (2,048 rays, 262,144 samples) rendered by the reference's Deformable.query_field + render_pixel, every 16th ray
    stored -- on W0 (train_bench.pt: raw seeded initialisation) and, round 5, on W1 (train_bench_w1.pt: the same weights after the reference's own
    geometry_init, SURVEY 8d: a fitted, sharp surface, where compositing weights are peaked).  Both precisions; plus the eval path's render on W1
    (eval_bench_w1.pt, first band: importance sampling, normals, compaction)."""
    import math
    from lab4d_amd import deformable as DF, mlp, synthetic

    def weights(meta):
        P = synthetic.make_weights(meta["seed"], sdf_bias=meta.get("sdf_bias"))
        if meta.get("w1"):  # the tensors geometry_init moved (tests/golden/w1_weights.pt; the same overlay tests/fixture_utils.fg_weights applies)
            P.update({k: v.clone() for k, v in torch.load(os.path.join(ROOT, "tests", "golden", "w1_weights.pt"), weights_only=False)["changed"].items()})
        return synthetic.to_device(P, dev)

    psnr = lambda a, b: round(-10.0 * math.log10(max(float(((a - b) ** 2).mean()), 1e-20)), 1)  # noqa: E731
    out = {}
    for tag, name in (("", "train_bench.pt"), ("w1_", "train_bench_w1.pt")):
        g = torch.load(os.path.join(ROOT, "tests", "golden", name), weights_only=False)
        meta = g["meta"]
        st, M, res, seed = meta["full_grid_stride"], meta["M"], meta["res"], meta["seed"]
        P = weights(meta)
        hxy = synthetic.make_rays(res, M, rows=meta.get("rows"))
        batch = synthetic.to_device(synthetic.make_targets(seed + 3, M, hxy.shape[1], res, hxy), dev)
        fr = synthetic.add_codes(synthetic.to_device(dict(g["frames"]), dev), P)
        fr["feature"] = batch["feature"]
        with torch.no_grad():
            for pname, prec in (("fp32", mlp.PREC_F32), ("bf16", mlp.PREC_BF16)):
                r = DF.render_train(P, fr, hxy.to(dev), synthetic.to_device(g["rng"], dev), flow_thresh=meta["flow_thresh"], n_depth=meta["D"],
                                    alpha=meta["alpha"], prec=prec)
                out[tag + pname] = psnr(r["rendered"]["rgb"][:, ::st].cpu(), g["rendered"]["rgb"])
        if not tag:
            out["case"] = "tests/golden/train_bench.pt (the reference's own render at the bench shape: %dx%d, %d samples/ray, %d rays rendered, every %dth compared)" \
                          % (res, res, meta["D"], hxy.shape[0] * hxy.shape[1], st)
    g = torch.load(os.path.join(ROOT, "tests", "golden", "eval_bench_w1.pt"), weights_only=False)
    meta = g["meta"]
    P = weights(meta)
    fr = synthetic.add_codes(synthetic.to_device(dict(g["frames"]), dev), P)
    hxy = synthetic.make_rays(meta["res"], meta["M"], rows=(meta["rows"][0], meta["rows"][0] + meta["band"]))
    with torch.no_grad():
        for pname, prec in (("fp32", mlp.PREC_F32), ("bf16", mlp.PREC_BF16)):
            r = DF.render_eval(P, fr, hxy.to(dev), n_depth=meta["D"], prec=prec)
            out["eval_w1_" + pname] = psnr(r["rendered"]["rgb"][:, ::meta["full_grid_stride"]].cpu(), g["rendered_bands"][0]["rgb"])
    out["case_w1"] = "train_bench_w1.pt / eval_bench_w1.pt: the same shapes on W1 = seed 61's weights after the reference's own geometry_init (500 Adam steps on the " \
                     "Gaussian-bone SDF); eval: importance sampling 64 + 64, first 2-row band"
    return out


class ExtrasBudget:
    """Wall-clock budget of everything the default run does BESIDE the headline (two rocprofv3 passes + three child benches + the CPU
    baseline used to run back to back with per-item timeouts only -- worst case > 25 min in front of the one JSON line).  Every extra asks for its
    timeout here: min(its own cap, what is left); an extra that finds less than `floor` seconds left is skipped and says so in the line."""

    def __init__(self, total_s):
        self.total = float(total_s)
        self.start()

    def start(self):
        """(Re)start the clock: called when the headline has been measured and the extras begin."""
        self.t_end = time.perf_counter() + self.total

    def left(self):
        return self.t_end - time.perf_counter()

    def timeout(self, cap, floor=30.0):
        left = self.left()
        return None if left < floor else min(float(cap), left)


EXTRAS = ExtrasBudget(float(os.environ.get("LAB4D_BENCH_EXTRAS_BUDGET_S", "900")))
SKIPPED = "skipped: the extras' wall-clock budget (LAB4D_BENCH_EXTRAS_BUDGET_S = %d s) was spent" % int(EXTRAS.total)


def fp32_leg(a):
    """The same training step with fp32 MFMA chains (`--dtype f32`: v_mfma_f32_32x32x2_f32, fp32 stored activations, 64-row chunks), 1 warm-up
    + 2 timed steps in a child process once this one has given its device memory back.  fp32 is the precision the reference computes in
    and the one the 1e-4 parity tests run: its throughput belongs next to the bf16 headline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--dtype", "f32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras",
           "--res", str(a.res), "--spp", str(a.spp)]
    tmo = EXTRAS.timeout(600)
    if tmo is None:
        return {"value": None, "error": SKIPPED}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=tmo, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {"value": d["value"], "unit": "rays/s", "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"], "dtype": "f32",
                "frac_of_fp32_mfma_peak": d["whole_graph_frac_of_peak"], "whole_graph_tflops": d["whole_graph_tflops"], "peak_hbm_gib": d["peak_hbm_gib"],
                "loss_last_chunk": d["loss_last_chunk"], "params_finite": d["params_finite"], "chunk_rays": d["config"]["chunk_rays"],
                "dominant_kernel": {k: d["roofline"][k] for k in ("kernel", "bound", "frac", "avg_ms")} if d.get("roofline") else None}
    except Exception as e:  # an extra: report the failure instead of losing the bench line
        return {"value": None, "error": repr(e)[:300]}


def other_configs_legs():
    """BASELINE configs[0]'s shape (64 x 64 x 64 through the fg loop) and configs[2] / [3] / [4] at their per-GPU shapes (--config comp | multi | hash), 2 warm-up + 5 timed steps each (50 for the small shape) in child processes after
    the headline run has given its device memory back: the driver-visible line carries every configuration BASELINE.json names, not only the fg one."""
    import subprocess
    out = {}
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for cfg in ("config0_shape", "comp", "multi", "hash"):
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--config", cfg, "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extras"]
        if cfg == "config0_shape":  # BASELINE configs[0]'s shape through the fg loop: a 64 x 64 crop of the frame pair, 64 samples/ray = 8,192 rays per step (launch-bound: 50 steps)
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--res", "64", "--spp", "64", "--chunk-rows", "64", "--steps", "50", "--warmup", "5", "--no-cpu-baseline", "--no-extras"]
        t0 = time.perf_counter()
        tmo = EXTRAS.timeout(420)
        if tmo is None:
            out[cfg] = {"value": None, "error": SKIPPED}
            continue
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=tmo, env=env)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            leg = {"value": d["value"], "unit": "rays/s", "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"], "dtype": d["dtype"],
                   "workload": d["config"]["workload"], "rays_per_step": d["config"]["rays_per_step"], "peak_hbm_gib": d.get("peak_hbm_gib"),
                   "loss_last_chunk": d.get("loss_last_chunk"), "params_finite": d.get("params_finite"), "wall_s": round(time.perf_counter() - t0, 1)}
            if d.get("whole_graph_frac_of_peak") is not None:
                leg["whole_graph_frac_of_peak"] = d["whole_graph_frac_of_peak"]
            if d.get("roofline"):
                leg["dominant_kernel"] = {k: d["roofline"][k] for k in ("kernel", "bound", "frac", "avg_ms")}
            if cfg == "hash":
                leg["parity"] = "unpinned (the reference has no hash grid: nnutils/nerf.py:98 is a comment)"
                leg["inside_box_fraction"] = d["config"].get("inside_box_fraction")
                top = sorted(((v["ms_per_step"], k) for k, v in d.get("kernels", {}).items()), reverse=True)[:3]
                leg["top_kernels_ms_per_step"] = {k: ms for ms, k in top}
            out[cfg] = leg
        except Exception as e:  # an extra: report the failure instead of losing the bench line
            out[cfg] = {"value": None, "error": repr(e)[:300], "wall_s": round(time.perf_counter() - t0, 1)}
    return out


def pmc_traffic_in_run():
    """HBM traffic of the chain / weight-gradient kernels measured IN THIS RUN (the line used to quote profiles/r04_pmc_traffic.json):
    rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (each with --kernel-trace only) over tools/bench_mlp.py at the bench's launch
    size, summarised by tools/pmc_summary.py with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE x 2; KiB units).  The
    passes profile the micro-bench, not this process (rocprofv3 --pmc around the whole bench.py has hung on this pool), each under its own timeout.
    Returns (summary dict or None, reason): the reason says why there is no measurement (no rocprofv3 on the box / LAB4D_BENCH_PMC=0 / which pass
    failed with which return code or timeout / the extras' budget) and goes into the line's `traffic_source`."""
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("LAB4D_BENCH_PMC", "1") == "0":
        return None, "LAB4D_BENCH_PMC=0"
    if shutil.which("rocprofv3") is None:
        return None, "no rocprofv3 on this box"
    tmp = tempfile.mkdtemp(prefix="lab4d_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["TMPDIR"] = "/tmp"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_mlp.py"), "16777216", "1"]
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            tmo = EXTRAS.timeout(120, floor=60.0)
            if tmo is None:
                return None, SKIPPED
            r = subprocess.run(["timeout", "-k", "5", str(int(max(tmo - 25, 30))), "rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, ctr), "--"] + cmd,
                               capture_output=True, text=True, cwd="/tmp", env=env, timeout=tmo)
            if r.returncode != 0:
                return None, "the rocprofv3 --pmc %s pass ended with return code %d: %s" % (ctr, r.returncode, (r.stderr or "")[-160:].replace("\n", " "))
        outp = os.path.join(tmp, "pmc.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(tmp, "FETCH_SIZE"), os.path.join(tmp, "WRITE_SIZE"), outp,
                            "measured in this bench.py run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/bench_mlp.py 16777216 1"],
                           capture_output=True, text=True, timeout=60)
        if r.returncode == 0 and os.path.exists(outp):
            return json.load(open(outp)), None
        return None, "tools/pmc_summary.py ended with return code %d: %s" % (r.returncode, (r.stderr or "")[-160:].replace("\n", " "))
    except Exception as e:  # (subprocess.TimeoutExpired included)
        return None, "pmc pass failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(res, spp, n_rays, config="fg"):
    """The oracle (CPU port of the reference algorithm) on a bounded sample of the same workload (same field configuration)."""
    from lab4d_amd import synthetic
    from oracle import lab4d_oracle as O
    # PyTorch's intra-op pool stops scaling (and collapses from oversubscription) well below the 256
    # hardware threads of the GPU host on these small per-sample ops; 32 threads measured best.
    threads = min(os.cpu_count(), 32)
    torch.set_num_threads(threads)
    leaf = lambda P: {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and k != "aabb" else v) for k, v in P.items()}  # noqa: E731
    Pb = frb0 = None
    if config == "comp":
        P = leaf(synthetic.add_dense_weights(synthetic.make_weights(0, num_bones=18), 0))
        fr0 = synthetic.make_frames(1, 2, res, num_bones=18)
        Pb = leaf(synthetic.make_bg_weights(0))
        frb0 = synthetic.make_bg_frames(1, 2, res)
    elif config == "multi":
        P = leaf(synthetic.add_dense_weights(synthetic.make_weights(0, num_inst=10), 0, num_inst=10))
        fr0 = synthetic.make_frames(1, 2, res, num_inst=10)
        fr0["inst_id"] = torch.full((2,), 3, dtype=torch.long)
    else:
        P = leaf(synthetic.make_weights(0))
        fr0 = synthetic.make_frames(1, 2, res)
    # memory guard: the oracle keeps the whole autograd graph of the sample (~5.6 MB per ray at 128 samples: 11.4 GB for 2 x 1,024 rays, measured)
    shrunk = ""
    try:
        avail = next(int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")) * 1024
        want = n_rays
        while n_rays > 64 and 2 * n_rays * 5.6e6 * (spp / 128.0) * 1.5 > avail:
            n_rays //= 2
        if n_rays != want:
            shrunk = " (asked for %d rays per frame; %.0f GB of host memory available)" % (want, avail / 1e9)
    except Exception:
        pass
    g = torch.Generator().manual_seed(0)
    hxy = torch.cat([torch.rand(2, n_rays, 2, generator=g) * res, torch.ones(2, n_rays, 1)], -1)
    batch = synthetic.make_targets(2, 2, n_rays, res, hxy)
    d_field = spp // 2 if config == "comp" else spp
    S = 2 * n_rays * d_field
    eik = torch.randperm(2 * n_rays, generator=g)[: max(2 * n_rays // 16, 1)]
    rng = {"eik_inds": eik, "eik_inds_bg": eik, "match_perm": torch.randperm(S, generator=g)[:1024]}

    def one_pass(h, b, r):
        f = synthetic.add_codes(dict(fr0), P)
        f["feature"] = b["feature"]
        if config == "comp":
            frb = synthetic.add_bg_codes(dict(frb0), Pb)
            out = O.render_train_comp(P, f, Pb, frb, h, r, flow_thresh=float(res), n_depth=d_field)
            sum(O.recon_losses_comp(out, b, res, O.DEFAULT_LOSS_WT).values()).backward()
        else:
            out = O.render_train(P, f, h, r, flow_thresh=float(res), n_depth=spp)
            sum(O.recon_losses_fg(out, b, res, O.DEFAULT_LOSS_WT).values()).backward()

    # SURVEY 8d protocol, bounded: 1 warm-up pass (a small sample: thread pool, allocator), then timed passes of the FULL sample until 20 s of timed
    # work or 3 passes, median.  The full sample is one chunk of the reference's own chunking of configs[1] (8,192 rays x 128 samples) when the
    # host has the memory for the oracle's autograd graph of it (~5.6 MB per ray, measured), else the largest halving that fits.
    nw = min(n_rays, 128)
    hxy_w = hxy[:, :nw].contiguous()
    one_pass(hxy_w, synthetic.make_targets(2, 2, nw, res, hxy_w),
             {"eik_inds": torch.arange(max(2 * nw // 16, 1)), "eik_inds_bg": torch.arange(max(2 * nw // 16, 1)), "match_perm": torch.randperm(2 * nw * d_field, generator=g)[:1024]})
    ts = []
    while len(ts) < 3 and (not ts or sum(ts) < 20.0):
        t0 = time.perf_counter()
        one_pass(hxy, batch, rng)
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    out = {"value": 2 * n_rays / dt, "unit": "rays/s", "cores": threads, "kind": "port",
           "sample": "oracle (torch-CPU fp32 port of the reference path, configuration %r) fwd+bwd, 2 frames x %d rays x %d samples%s, 1 small warm-up + median of %d "
                     "timed pass(es) (%.1f s each) on %d threads of the GPU box's host" % (config, n_rays, spp, shrunk, len(ts), dt, threads)}
    ref = os.path.join(ROOT, "profiles", "r02_cpu_reference.json")
    if config == "fg" and os.path.exists(ref):  # the REFERENCE's own code timed in the build container (it cannot run on the GPU box): cited, not re-measured
        out["reference_in_build_container"] = json.load(open(ref))
    return out


def rank_rows(rank, world, res):
    """Strong scaling: rank r renders image rows r, r + world, r + 2 world, ... of both frames (round-robin: every rank sees the whole
    image; the first res % world ranks hold one row more)."""
    return list(range(rank, res, world))


def chunk_rows_of(rows, n_chunks):
    """A rank's rows dealt out to its chunks round-robin (chunk c takes rows[c::n_chunks])."""
    return [rows[c::n_chunks] for c in range(n_chunks)]


def allreduce_flat(flat_grad, world):
    """THE collective of the data-parallel job: one all-reduce (RCCL on the GPUs, gloo in the CPU tests) of the optimizer's flat
    fp32 gradient buffer, averaged -- the reference's DDP semantics (per-rank loss normalisers, mean over ranks)."""
    dist.all_reduce(flat_grad)
    flat_grad.div_(world)


def rank_plan(rank, world, res, chunk_rows, spp):
    """Rows, chunk list and device-memory estimate of one rank (also what --dry-ranks prints for every rank)."""
    rows = rank_rows(rank, world, res)
    n_chunks = max(1, -(-len(rows) // chunk_rows))
    chunks = chunk_rows_of(rows, n_chunks)
    samples = max(2 * len(c) * res * spp for c in chunks)
    return {"rank": rank, "rows": "%d::%d (%d rows)" % (rank, world, len(rows)), "n_rows": len(rows), "chunks": chunks,
            "chunk_sizes": [len(c) for c in chunks], "uniform": len({len(c) for c in chunks}) == 1,
            "rays_per_step": 2 * res * len(rows), "peak_chunk_samples": samples,
            "est_peak_hbm_gib": round(samples * BYTES_PER_SAMPLE / 2**30, 1)}


def dry_ranks(a):
    """De-risk --gpus N without the hardware: every rank's plan, and the invariants the multi-process run relies on."""
    world = a.dry_ranks
    plans = [rank_plan(r, world, a.res, a.chunk_rows, a.spp) for r in range(world)]
    rows = sorted(y for p in plans for c in p["chunks"] for y in c)
    assert rows == list(range(a.res)), "the ranks' chunks must partition the rows of the frame"
    assert sum(p["rays_per_step"] for p in plans) == 2 * a.res * a.res
    for p in plans:
        p["launch"] = "hipGraph replay per chunk" if p["uniform"] else "eager (chunk_rows does not divide the rank's rows: chunks differ in shape)"
        assert p["est_peak_hbm_gib"] < 250, "rank %d would not fit 288 GB" % p["rank"]
        p["chunks"] = ["%d::%d" % (c[0], c[1] - c[0]) if len(c) > 1 else str(c) for c in p["chunks"]]  # first row :: stride
    print(json.dumps({"world": world, "collective": "1 x all_reduce of the flat fp32 gradient per step", "plans": plans}))


def hash_main(a):
    """BASELINE configs[4]'s per-GPU shape: rays -> samples -> hash-grid field (lab4d_amd/hashfield.py) -> compositing -> colour + mask loss ->
    backward -> AdamW, one 1024^2 frame pair x 256 samples/ray per step in row-interleaved chunks.  No reference counterpart (the reference
    has no hash grid): an absolute number for the encoding variant north_star names, not a parity claim."""
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from lab4d_amd import _lib, hashfield, mlp, synthetic
    from lab4d_amd import quat_utils as Q, render_utils as RU
    from lab4d_amd.optim import FlatAdamW
    _lib.lib()
    res = a.res if a.res_given else 1024
    spp = a.spp if a.spp_given else 256
    rows = a.chunk_rows if a.chunk_rows_given else 16
    prec = mlp.PREC_BF16 if a.dtype == "bf16" else mlp.PREC_F32
    P, cfg = hashfield.make_weights(0, sdf_bias=0.02)
    P = synthetic.to_device(P, dev)
    params = [v for k, v in P.items() if k != "aabb"]
    for v in params:
        v.requires_grad_(True)
    fr = synthetic.to_device(synthetic.make_frames(1, 2, res), dev)
    cam2field = Q.quaternion_translation_inverse(fr["field2cam"][0], fr["field2cam"][1])
    hres = hashfield.resolutions(cfg, dev)
    opt = FlatAdamW(params, lr=1e-3)
    mlp.FUSED_GRAD_ACCUM = True
    n_chunks = res // rows
    chunks = chunk_rows_of(list(range(res)), n_chunks)
    inputs = []
    for i, rr in enumerate(chunks):
        hxy = synthetic.make_rays(res, 2, rows=rr).to(dev)
        g = torch.Generator().manual_seed(100 + i)
        tgt = {"rgb": torch.rand(2, hxy.shape[1], 3, generator=g).to(dev), "mask": ((hxy[..., :2] - res / 2).norm(dim=-1, keepdim=True) < res / 4).float()}
        inputs.append((hxy, tgt))
    M, N0 = inputs[0][0].shape[:2]
    # Round 6: the field is evaluated on the samples inside the box only (hashfield.forward_compacted: device-side stream compaction into a buffer of
    # STATIC capacity, the chunk is a captured graph).  The capacity is set from the chunks' own inside counts (the rays are the bench's fixed pixel grid;
    # a training loader's random rays would size it from the box / frustum geometry the same way) with a 25 % margin; the device-side overflow flags
    # are checked behind the run -- a dropped sample fails the leg, it does not flatter it.
    S_chunk = M * N0 * spp
    counts = []
    with torch.no_grad():
        for hxy, _ in inputs:
            xyz = RU.ray_samples(hxy, fr["Kinv"], fr["near_far"], cam2field, n_depth=spp)[4].reshape(-1, 3)
            x01 = (xyz - P["aabb"][0]) / (P["aabb"][1] - P["aabb"][0])
            counts.append(int(((x01 >= 0) & (x01 <= 1)).all(-1).sum()))
    cap = min(S_chunk, (int(1.25 * max(counts)) + 1023) // 1024 * 1024) if not a.hash_no_compact else None
    overflow_any = torch.zeros(1, dtype=torch.bool, device=dev)
    f16g = not a.hash_f32_table_grad

    def chunk(hxy, tgt):
        _, _, deltas, _, xyz, dirs = RU.ray_samples(hxy, fr["Kinv"], fr["near_far"], cam2field, n_depth=spp)
        if cap is None:
            rgb, dens = hashfield.forward(P, cfg, xyz.reshape(-1, 3), dirs.reshape(-1, 3), spf=N0 * spp, prec=prec, res=hres, table_grad_f16=f16g)
        else:
            rgb, dens, _, ovf = hashfield.forward_compacted(P, cfg, xyz.reshape(-1, 3), dirs.reshape(-1, 3), cap, prec=prec, res=hres, table_grad_f16=f16g)
            overflow_any.logical_or_(ovf)
        r = RU.render_pixel({"rgb": rgb.view(M, N0, spp, 3), "density": dens.view(M, N0, spp, 1)}, deltas)
        loss = (r["rgb"] - tgt["rgb"]).pow(2).mean() + 0.1 * (r["mask"] - tgt["mask"]).pow(2).mean()
        loss.backward()
        return loss.detach()

    st_hxy, st_tgt = inputs[0][0].clone(), {k: v.clone() for k, v in inputs[0][1].items()}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            chunk(st_hxy, st_tgt)
    torch.cuda.current_stream().wait_stream(side)
    graph = None
    if not a.no_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            st_loss = chunk(st_hxy, st_tgt)
    opt.zero_grad()

    def step():
        opt.zero_grad()
        last = None
        for hxy, tgt in inputs:
            if graph is not None:
                st_hxy.copy_(hxy)
                for k in tgt:
                    st_tgt[k].copy_(tgt[k])
                graph.replay()
                last = st_loss
            else:
                last = chunk(hxy, tgt)
        opt.step(max_norm=5.0)
        mlp.repack_all()
        return last

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.PROF = {}
    for hxy, tgt in inputs[:4]:
        chunk(hxy, tgt)
    torch.cuda.synchronize()
    prof = _lib.prof_summary()
    _lib.PROF = None
    rays = 2 * res * res
    value = rays * a.steps / dt
    n_in, n_all = sum(counts), S_chunk * len(inputs)  # how much of the step carries a field at all: the grid is defined on the box only
    if bool(overflow_any):
        fail("--config hash: a chunk held more inside-box samples than the compaction buffer (%d rows): samples were dropped" % cap)
    kern = {k: {"ms_per_step": round(v[1] / 4 * len(inputs), 2), "GBps": round(v[3] / v[1] / 1e6, 1) if v[3] else None} for k, v in sorted(prof.items())}
    # roofline of the leg's dominant kernel (by event-measured time over the 4 profiled chunks): the table gradient is bound by the L2's atomic rate,
    # not by bytes -- both are stated: algorithmic bytes (2 x 8 vertices x F floats per level read / added, the point, the encoding gradient row) against
    # 8 TB/s, and scalar fp32 atomic adds per second (inside samples x L x 8 x F; runs of equal vertices across a wave are combined first, so fewer reach the L2)
    roof = None
    ranked = sorted(((k, v) for k, v in prof.items() if v[3] > 0), key=lambda kv: -kv[1][1])
    if ranked:
        name, (launches, ms, _, nbytes) = ranked[0]
        roof = {"bound": "hbm", "kernel": name, "achieved": round(nbytes / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / ms / 1e6 / 8000.0, 4),
                "traffic": None, "launches": launches, "avg_ms": round(ms / launches, 4), "algorithmic_per_launch": {"bytes": nbytes / launches},
                "measured": "HIP events around every launch in an eager re-run of 4 chunks right after the timed region"}
        if name == "k_hashgrid_bwd":
            roof["atomic_adds_per_s_upper"] = round(sum(counts[:4]) * cfg["L"] * 8 * cfg["F"] / (ms * 1e-3), 0)
            roof["note"] = "bound by the L2's fp32 atomic-add rate, not by bytes: the fraction of the HBM roof is reported for the contract, the atomic rate is what the kernel sits at"
    out = {"metric": "rendered rays/sec (fwd+bwd), hash-grid field at %dx%d x %d samples (BASELINE configs[4] per-GPU shape; no reference counterpart, not the headline metric)" % (res, res, spp),
           "value": round(value, 1), "unit": "rays/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": "hash-grid field (L=16, F=2, T=2^19, 16..2048; geometry net 32-64-16, colour net 19-64-64-3), %dx%d frame pair, %d samples/ray, "
                                  "rays -> samples -> field -> compositing -> rgb + mask loss -> backward -> AdamW" % (res, res, spp),
                      "rays_per_step": rays, "chunk_rays": 2 * rows * res, "launch": "hipGraph replay per chunk" if graph is not None else "eager",
                      "inside_box_fraction": round(n_in / max(n_all, 1), 4),
                      "field_rows_per_chunk": (cap if cap is not None else S_chunk), "samples_per_chunk": S_chunk,
                      "table_gradient": ("dense levels: fp32 atomics with wave-level run combining; hashed levels: ONE packed 2 x fp16 atomic per vertex at a per-launch "
                                         "power-of-two scale (Instant-NGP's fp16 gradient accumulation), flushed into the fp32 gradient after every chunk") if f16g
                                        else "fp32 atomics on every level (--hash-f32-table-grad)",
                      "compaction": ("the field (encoding, both nets, their weight gradients, the table gradient) runs on the inside-box samples only: device-side "
                                     "stream compaction into %d rows per chunk (max inside count %d + 25 %%), overflow checked" % (cap, max(counts))) if cap is not None else "off (--hash-no-compact)",
                      "field_support": "the box only (Instant-NGP 5.4): samples outside carry no density / colour and no table gradient -- not comparable with the "
                                       "rounds 1-3 numbers of this leg, which clamped them onto the boundary cells",
                      "parity": "unpinned: the reference has no hash grid"},
           "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2**30, 1), "kernels": kern, "roofline": roof,
           "loss_last_chunk": float(last), "params_finite": bool(all(bool(torch.isfinite(p).all()) for p in params))}
    sys.stdout.flush()
    real_stdout.write(json.dumps(out) + "\n")
    real_stdout.flush()


def fail(msg, code=2):
    print("bench.py: " + msg, file=sys.stderr)
    sys.exit(code)


def launch_ranks(a):
    """`python bench.py --gpus N` with no launcher in front: start the N ranks here, one process per GPU, the way the reference starts its own
    (scripts/train.sh:12-16 `torchrun --nproc_per_node $ngpu`, lab4d/train.py:28-33), and hand rank 0's JSON line through.  Refuses loudly when the
    box has fewer than N GPUs -- a 1-GPU number labelled as an N-GPU one is worse than no number."""
    import socket
    import subprocess
    if a.backend == "nccl" and not a.dry_step:
        found = torch.cuda.device_count()
        if found < a.gpus:
            fail("--gpus %d needs %d GPUs, found %d" % (a.gpus, a.gpus, found))
    if a.share_gpu and torch.cuda.device_count() < 1:
        fail("--share-gpu needs at least one GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    sys.exit(subprocess.run(cmd, env=env).returncode)


def dry_step(a):
    """--dry-step: the multi-process skeleton of a step without a GPU.  Every rank: its row plan, the product's bucket (FlatAdamW.flat_grad over the fg
    parameter set) filled with a rank-dependent gradient, the step's ONE collective (allreduce_flat), barrier + max-over-ranks timing as in the real
    run; rank 0 prints the line with n_gpus = the ranks the process group actually saw."""
    from lab4d_amd import synthetic
    from lab4d_amd.optim import FlatAdamW
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")  # gloo announces its connections on fd 1: stdout proper carries the JSON line only (as in rank_main)
    os.dup2(2, 1)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.set_num_threads(1)
    dist.init_process_group(a.backend, rank=rank, world_size=world)
    seen = dist.get_world_size()
    plan = rank_plan(rank, world, a.res, a.chunk_rows, a.spp)
    P = synthetic.make_weights(0)
    params = [v.requires_grad_(True) for k, v in P.items() if v.dtype.is_floating_point and k != "aabb"]
    opt = FlatAdamW(params, lr=5e-4)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        opt.zero_grad()
        for p in params:
            p.grad.fill_(float(rank + 1))  # stands for the rank's accumulated chunk gradients (views of the flat bucket)
        allreduce_flat(opt.flat_grad, world)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0, float(plan["rays_per_step"])], dtype=torch.float64)
    ts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(ts, t)
    ok = bool(torch.allclose(params[0].grad, torch.full_like(params[0].grad, (world + 1) / 2.0)))
    dist.destroy_process_group()
    if rank == 0:
        dt = max(float(x[0]) for x in ts)
        real_stdout.write(json.dumps({"metric": "dry step (no GPU work): launch + rendezvous + the flat-gradient all-reduce of bench.py --gpus N", "value": 0.0, "unit": "rays/s",
                          "n_gpus": seen, "ranks_seen_by_process_group": seen, "backend": a.backend, "dry_step": True, "steps": a.steps, "warmup": 0,
                          "ms_per_step": round(dt / max(a.steps, 1) * 1e3, 3), "rays_per_step_all_ranks": int(sum(float(x[1]) for x in ts)),
                          "allreduce_is_rank_mean": ok, "bucket_elements": int(opt.n)}) + "\n")
        real_stdout.flush()
    if not ok:
        sys.exit(3)


def main():
    a = parse()
    if a.dry_ranks:
        return dry_ranks(a)
    env_world = os.environ.get("WORLD_SIZE")
    if a.gpus is None:
        a.gpus = int(env_world) if env_world else 1
    if a.gpus < 1:
        fail("--gpus must be >= 1")
    if a.backend == "gloo" and not (a.dry_step or a.share_gpu):
        fail("--backend gloo runs no GPU work: use it with --dry-step, or with --share-gpu for a functional multi-rank run on fewer GPUs than ranks")
    if a.share_gpu and a.backend != "gloo":
        fail("--share-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
    if env_world is None and a.gpus > 1:
        return launch_ranks(a)  # does not return
    if env_world is not None and int(env_world) != a.gpus:
        fail("--gpus %d under a launcher that started %s rank(s): the two must agree" % (a.gpus, env_world))
    if a.dry_step:
        return dry_step(a)
    if a.config == "hash":
        if a.gpus > 1:
            fail("--config hash is a one-GPU leg")
        return hash_main(a)
    rank_main(a)


class TrainLoop:
    """The benchmark's training step as an object: problem + resident inputs + per-frame prologue + (optionally) the captured chunk graph +
    FlatAdamW.  `step()` = zero_grad -> prologue -> every chunk (hipGraph replay or eager) -> prologue backward -> [all-reduce] ->
    check_grad + AdamW -> repack.  bench.py times it; tests/test_gpu_ztrajectory.py runs it for 30+ steps."""

    def __init__(self, dev, res, spp, chunks, prec, comp=False, use_graph=True, use_dist=False, world=1, rank=0, trace=False, lr=5e-4, multi=False,
                 step_graph=False):
        from lab4d_amd import mlp
        from lab4d_amd import deformable as DF
        from lab4d_amd.optim import FlatAdamW
        self.DF, self.mlp = DF, mlp
        self.dev, self.res, self.spp, self.prec, self.comp, self.use_dist, self.world, self.trace_on = dev, res, spp, prec, comp, use_dist, world, trace
        Pb = frb = None
        if comp:
            P, fr, Pb, frb = make_problem(res, dev, comp=True)
        else:
            P, fr = make_problem(res, dev, multi=multi)
        self.P, self.Pb = P, Pb
        self.params = [v for k, v in P.items() if v.dtype.is_floating_point and v.requires_grad] + (list(Pb.values()) if comp else [])
        self.param_names = [k for k, v in P.items() if v.dtype.is_floating_point and v.requires_grad] + (["bg." + k for k in Pb] if comp else [])
        # the reference's optimizer step (trainer.py:164-190,349-350,581-604): check_grad (clip_grad_norm_(params, 5.0), discard above it) + AdamW,
        # one learning rate per parameter -- here three launches over one flat buffer; p.grad are views of opt.flat_grad, which is also the
        # all-reduce bucket
        self.opt = FlatAdamW(self.params, lr=lr)
        mlp.FUSED_GRAD_ACCUM = True  # weight-gradient kernels add straight into those views (no scatter / AccumulateGrad per layer)
        self.gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        # pre-build the inputs (resident in HBM before the timed region)
        self.inputs = [chunk_inputs(res, None, rows, dev, seed=100 + i) for i, rows in enumerate(chunks)]
        self.fr0 = fr
        # per-frame prologue: camera inverses, bone transforms, per-frame bias tables -- functions of (weights, frames) only, evaluated
        # once per step; the chunks read them from static leaves and the summed leaf gradients go back through it after the last chunk
        self.prologue = DF.FramePrologue(P, fr)
        self.fr = self.prologue.refresh()
        self.prologue.outs = None  # no autograd graph of the prologue (and none of its AccumulateGrad nodes) alive while the chunk is captured
        self.prologue_bg = None
        self.frb = None
        if comp:
            self.prologue_bg = DF.BgPrologue(Pb, frb)
            self.frb = self.prologue_bg.refresh()
            self.prologue_bg.outs = None
        self.M, self.N0 = self.inputs[0][0].shape[:2]
        self.S0 = self.M * self.N0 * (spp // 2 if comp else spp)
        self.uniform = all(h.shape == self.inputs[0][0].shape for h, _ in self.inputs)
        self.graph = None
        self.graph_a = self.graph_b = None
        self.ar_events = []
        if use_graph and self.uniform and step_graph and not trace:
            try:
                self.capture_step()
            except Exception as e:  # a failed whole-step capture must not cost the run: fall back to the per-chunk graphs (the line's `launch` says which ran)
                print("bench.py: whole-step graph capture failed (%r); falling back to per-chunk graphs" % (e,), file=sys.stderr)
                self.graph_a = self.graph_b = None
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                self.capture()
        elif use_graph and self.uniform:
            self.capture()
        self.opt.zero_grad()
        self.prologue.zero_grad()  # the eager warm-up / capture passes accumulated into the leaves
        if comp:
            self.prologue_bg.zero_grad()

    def chunk(self, hxy, batch, rng):
        if self.comp:
            return train_chunk_comp(self.DF, self.P, self.fr, self.Pb, self.frb, hxy, batch, rng, self.spp, self.res, self.prec)
        return train_chunk(self.DF, self.P, self.fr, hxy, batch, rng, self.spp, self.res, self.prec)

    def capture(self):
        # One chunk (forward + losses + backward, ~4000 launches) is captured once as a hipGraph and replayed for every
        # chunk: inputs are copied into static buffers, gradients accumulate in place in opt.flat_grad.  The packed bf16
        # copies of the weights live in persistent buffers the graph reads; they are refreshed in place once per step, after
        # the optimizer (mlp.repack_all) -- not once per chunk inside the graph.
        self.st_hxy = self.inputs[0][0].clone()
        self.st_batch = {k: v.clone() for k, v in self.inputs[0][1].items()}
        self.st_batch["hxy"] = self.st_hxy
        self.st_rng = draw_rng(self.M, self.N0, self.S0, self.dev, self.gen)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):  # eager warm-up on the capture stream: allocator pools, rocBLAS workspaces, column maps, packed weights
                self.chunk(self.st_hxy, self.st_batch, self.st_rng)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()  # the eager warm-up's blocks go back to the device: the graph's private pool needs the same amount again
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.st_loss = self.chunk(self.st_hxy, self.st_batch, self.st_rng)

    def capture_step(self):
        """Round 5 (VERDICT r04 "next" 6): the optimizer step as TWO hipGraphs instead of one graph per chunk + ~400 eager launches --
        graph A = zero the flat gradient, prologue refresh, EVERY chunk (each reads its own resident inputs and its own static random draws: no
        copies), prologue backward; [the data-parallel all-reduce, eager, between the two]; graph B = check_grad + AdamW (device-side discard rule and
        step count: optim.py) + the in-place repack of the kernels' weight copies.  What stays eager per step: the chunks' random draws, batched
        (draw_rng_all: 61 kernel launches + 3 copies, tools/count_step_launches.py; the per-chunk graphs needed 344 + 88 + 4 replays).  Measured
        time-neutral (profiles/r05_ab_*graph*.json): the eager launches were already hidden behind the running chunk.  Memory: the chunks are captured one behind the other on one stream, so a chunk's activations are freed into the
        graph's pool before the next chunk allocates -- the pool peaks at one chunk, like the per-chunk graph."""
        for hxy, batch in self.inputs:
            batch["hxy"] = hxy
        first = [draw_rng(self.M, self.N0, self.S0, self.dev, self.gen) for _ in self.inputs]
        self.rng_stack = (torch.stack([r["eik_inds"] for r in first]).contiguous(), torch.stack([r["match_perm"] for r in first]).contiguous())
        self.st_rngs = [{"eik_inds": self.rng_stack[0][c], "match_perm": self.rng_stack[1][c]} for c in range(len(first))]  # row views: static addresses

        def body_a():
            self.opt.flat_grad.zero_()  # (a fill kernel; the HIP runtime's memset NODES are what DESIGN.md section 2, finding 4 is about)
            self.prologue.refresh()
            if self.comp:
                self.prologue_bg.refresh()
            losses = [self.chunk(hxy, batch, rng) for (hxy, batch), rng in zip(self.inputs, self.st_rngs)]
            self.prologue.backward()
            if self.comp:
                self.prologue_bg.backward()
            return losses

        def body_b():
            self.opt.step(max_norm=5.0, skip_above=GRAD_SKIP)
            self.mlp.repack_all()

        # The eager warm-up (allocator pools, column maps, packed weights) runs one whole step -- and must NOT leave a parameter update behind: in a
        # --gpus N run the ranks render different rows, there is no all-reduce inside the warm-up, so a kept update would de-synchronise the
        # replicas for the rest of the run; on one GPU it would be a step the loss trajectory does not count.  The optimizer's
        # whole state is snapshotted in front of it and restored behind it, then the kernels' weight copies are re-packed from the restored weights.
        opt = self.opt
        snap = [t.clone() for t in (opt.flat, opt.m, opt.v, opt.dev_step, opt.skipped, opt.norm, opt.coef)]
        steps0 = opt.steps
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            body_a()
            body_b()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for t, s in zip((opt.flat, opt.m, opt.v, opt.dev_step, opt.skipped, opt.norm, opt.coef), snap):
                t.copy_(s)
        opt.steps = steps0
        torch.autograd.graph.increment_version(opt.params)
        self.mlp.repack_all()
        snap = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_a):
            self.st_losses = body_a()
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool()):
            body_b()
        self.opt.steps -= 1  # (capturing body_b counted a step on the host that the device did not take)
        self.st_loss = self.st_losses[-1]

    def release_graph(self):
        self.st_loss = self.st_losses = None
        self.graph = self.graph_a = self.graph_b = None
        torch.cuda.empty_cache()

    def trace(self, what, t=None):
        """--trace: synchronise and report (stderr).  t: a loss vector, or None for the state of the gradient / parameter buffers."""
        torch.cuda.synchronize()
        if t is not None:
            print("[trace] %s " % what + " ".join("%s=%.4g" % (k, float(x)) for k, x in zip(self.DF.LOSS_TERMS + ["total"], t)), file=sys.stderr)
            return
        opt, params, param_names = self.opt, self.params, self.param_names
        bad_g = [n for n, q in zip(param_names, params) if not bool(torch.isfinite(q.grad).all())]
        bad_p = [n for n, q in zip(param_names, params) if not bool(torch.isfinite(q).all())]
        top = sorted(((float(q.grad.abs().max()), n) for n, q in zip(param_names, params)), reverse=True)[:4]
        print("[trace] %s grad_norm=%.6g coef=%.4g skipped=%d |p|max=%.4g largest |grad|: %s non-finite grads: %s params: %s"
              % (what, float(opt.norm), float(opt.coef), int(opt.skipped), float(opt.flat.abs().max()), ["%s=%.3g" % (n, v) for v, n in top], bad_g[:6],
                 bad_p[:6]), file=sys.stderr)

    def step(self):
        opt, comp = self.opt, self.comp
        if self.graph_a is not None:
            draw_rng_all(len(self.inputs), self.M, self.N0, self.S0, self.dev, self.gen, self.rng_stack)
            self.graph_a.replay()
            if self.use_dist:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                allreduce_flat(opt.flat_grad, self.world)
                e1.record()
                self.ar_events.append((e0, e1))
            self.graph_b.replay()
            opt.steps += 1
            return self.st_loss
        opt.zero_grad()
        last = None
        self.prologue.refresh()
        if comp:
            self.prologue_bg.refresh()
        for ci_, (hxy, batch) in enumerate(self.inputs):
            if self.graph is not None:
                self.st_hxy.copy_(hxy)
                for k, v in batch.items():
                    if k != "hxy":
                        self.st_batch[k].copy_(v)
                draw_rng(self.M, self.N0, self.S0, self.dev, self.gen, out=self.st_rng)
                self.graph.replay()
                last = self.st_loss
            else:
                S = hxy.shape[0] * hxy.shape[1] * (self.spp // 2 if comp else self.spp)
                last = self.chunk(hxy, batch, draw_rng(hxy.shape[0], hxy.shape[1], S, self.dev, self.gen))
            if self.trace_on:
                self.trace("step %d chunk %d" % (opt.steps, ci_), last)
        self.prologue.backward()
        if comp:
            self.prologue_bg.backward()
        if self.use_dist:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            allreduce_flat(opt.flat_grad, self.world)
            e1.record()
            self.ar_events.append((e0, e1))
        opt.step(max_norm=5.0, skip_above=GRAD_SKIP)
        self.mlp.repack_all()
        if self.trace_on:
            self.trace("step %d" % (opt.steps - 1))
        return last


def graph_ok(loop, graph):
    """The sustained leg rides on the captured graphs (eager steps would time launch overhead, not the kernels)."""
    return loop.graph_a is not None or graph is not None


def rank_main(a):
    # stdout carries exactly ONE line, the JSON.  Native libraries write banners to file descriptor 1 (RCCL prints its version block there
    # on first use): fd 1 is pointed at stderr for the run, the JSON goes to a private duplicate of the original stdout at the very end.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if a.share_gpu:
        local = local % max(torch.cuda.device_count(), 1)
    if a.poison:
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True
    use_dist = world > 1 or a.force_dist
    if torch.cuda.device_count() <= local:
        fail("rank %d (local rank %d) has no GPU: %d visible on this box" % (rank, local, torch.cuda.device_count()))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        if a.share_gpu:
            dist.init_process_group("gloo")  # (CUDA tensors are staged through the host: functional, not fast)
        else:
            import datetime
            # (a collective that cannot complete -- a rank that died, a fabric that is not up -- aborts after 5 minutes instead of holding the node until the driver's limit)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=300))  # RCCL on ROCm, bound to this rank's GPU
        if dist.get_world_size() != a.gpus and not a.force_dist:
            fail("the process group holds %d rank(s), --gpus says %d" % (dist.get_world_size(), a.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if use_dist:
        # the communicator is built lazily at the first collective: build it NOW, before any hipGraph is captured and before the training loop's
        # private pools exist (a first-use setup of RCCL's buffers / IPC handles behind 150 GiB of captured pools is one more thing an 8-GPU box
        # would meet for the first time); also a loud early failure when the ranks cannot reach each other
        probe = torch.full((1,), float(rank + 1), device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        if abs(float(probe) - world * (world + 1) / 2.0) > 1e-3:
            fail("rank %d: the process group's first all-reduce returned %g, expected %g" % (rank, float(probe), world * (world + 1) / 2.0))

    from lab4d_amd import _lib, mlp
    from lab4d_amd import deformable as DF
    _lib.lib()
    prec = mlp.PREC_BF16 if a.dtype == "bf16" else mlp.PREC_F32
    res, spp = a.res, a.spp
    comp, multi = a.config == "comp", a.config == "multi"

    # strong scaling: this rank renders rows rank::world of both frames, its chunks interleave those rows again
    plan = rank_plan(rank, world, res, a.chunk_rows, spp) if not a.emulate_rank_of else rank_plan(0, a.emulate_rank_of, res, a.chunk_rows, spp)
    rays_per_step = 2 * res * res if not a.emulate_rank_of else plan["rays_per_step"]

    # SURVEY 8d also asks for the forward-only rate: eval-mode render (importance sampling -> 128 samples, field, normals
    # through one first-order backward, compositing) of the same rays.  Measured before the training graph is captured
    # (its 150 GiB private pool would leave the allocator thrashing), half a training chunk per call.
    eval_result = None
    if world == 1 and rank == 0 and not comp and not multi and not a.trace and not a.no_extras:
        try:
            P_e, fr_e = make_problem(res, dev)
            inputs_e = [chunk_inputs(res, None, rows, dev, seed=100 + i) for i, rows in enumerate(plan["chunks"])]
            eval_result = eval_rate(DF, P_e, fr_e, inputs_e, spp, prec, not a.no_graph)
        except Exception as e:  # an extra, never the headline: report the failure instead of losing the bench line
            eval_result = {"value": None, "error": repr(e)[:300]}
        P_e = fr_e = inputs_e = None
        mlp.clear_caches()
        torch.cuda.empty_cache()

    loop = TrainLoop(dev, res, spp, plan["chunks"], prec, comp=comp, use_graph=not a.no_graph, use_dist=use_dist, world=world, rank=rank, trace=a.trace,
                     multi=multi, step_graph=not a.chunk_graph)  # (round 6: comp too -- the bg prologue's backward is an explicit index_add_, deformable.BgPrologue.backward)
    opt, params, inputs, step = loop.opt, loop.params, loop.inputs, loop.step
    M, N0, S0, gen = loop.M, loop.N0, loop.S0, loop.gen
    graph = loop.graph if loop.graph is not None else loop.graph_a
    ar_events = loop.ar_events

    for _ in range(a.warmup):
        step()

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if graph is None:
        _lib.PROF = {}
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = step()
    sync()
    dt = time.perf_counter() - t0
    prof_src = "HIP events around every launch of the family during the timed steps"
    n_prof_chunks = len(inputs) * a.steps
    loss_last = float(last[12])
    peak_hbm = torch.cuda.max_memory_allocated()
    # The package sits at its power cap under the chain kernels: the longer the run, the warmer the part and the lower the sustained clock (DESIGN.md
    # section 5: 25-step runs 695-705 ms, a 200-step run 715 ms on the same kernels).  The headline is the K steps the contract times; the sustained rate is
    # reported BESIDE it: the same loop continued for --sustain-steps more steps (default single-GPU fg run only), timed on its own.
    sustained = None
    n_sus = a.sustain_steps if (world == 1 and a.config == "fg" and not a.no_extras and not a.emulate_rank_of and not a.trace and graph_ok(loop, graph)) else 0
    if n_sus > 0:
        sync()
        ts0 = time.perf_counter()
        for _ in range(n_sus):
            last_s = step()
        sync()
        dts = time.perf_counter() - ts0
        sustained = {"value": round(rays_per_step * n_sus / dts, 1), "unit": "rays/s", "ms_per_step": round(dts / n_sus * 1e3, 2), "steps": n_sus,
                     "after_steps": a.warmup + a.steps, "loss_last_chunk": float(last_s[12]),
                     "note": "the same loop continued behind the timed region (steps %d..%d of the run): the rate a long run settles at under the power cap"
                             % (a.warmup + a.steps + 1, a.warmup + a.steps + n_sus)}
    launch_mode = ("two hipGraph replays per optimizer step (all chunks + prologue | check_grad + AdamW + repack)" if loop.graph_a is not None else
                   "hipGraph replay per chunk" if graph is not None else "eager")
    n_skipped = int(opt.steps - int(opt.dev_step))  # steps check_grad discarded (0 on a healthy run)
    if graph is not None:
        # the eager re-run below needs the memory the graph's private pool holds
        last = graph = None
        loop.release_graph()
        graph_was = True
    else:
        graph_was = False
    if graph_was:
        # a replayed hipGraph cannot host HIP events between its launches: the per-family durations come from an eager
        # re-run of 4 chunks right after the timed region (same kernels, same sizes, same stream)
        _lib.PROF = {}
        n_prof_chunks = min(4, len(inputs))
        for hxy, batch in inputs[:n_prof_chunks]:
            loop.chunk(hxy, batch, draw_rng(M, N0, S0, dev, gen))
        torch.cuda.synchronize()
        prof_src = "HIP events around every launch of the family in an eager re-run of %d chunks right after the timed region " \
                   "(the timed region replays a captured hipGraph, which cannot host events)" % n_prof_chunks
    prof = _lib.prof_summary()
    _lib.PROF = None
    rank_ms = [dt / a.steps * 1e3]
    allreduce_ms = None
    if use_dist:
        t = torch.tensor([dt], device=dev)
        ts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ts, t)
        rank_ms = [float(x) / a.steps * 1e3 for x in ts]
        dt = max(float(x) for x in ts)  # the job is as slow as its slowest rank
        timed_ar = ar_events[-a.steps:]
        allreduce_ms = sum(e0.elapsed_time(e1) for e0, e1 in timed_ar) / max(len(timed_ar), 1)
    # data-parallel invariant: every replica holds the SAME weights after the same steps (identical initialisation, identical all-reduced gradients,
    # identical optimizer arithmetic) -- bit for bit.  Three checksums of the flat parameter buffer, gathered and compared (a replica that took a local
    # step the others did not -- the round-5 capture warm-up did -- shows up here).
    replicas_identical = None
    if use_dist:
        wv = opt.flat
        chk = torch.stack([wv.double().sum(), wv.double().abs().sum(), wv.view(torch.int32).long().sum().double()])
        chks = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(chks, chk)
        replicas_identical = bool(all(torch.equal(c, chks[0]) for c in chks))
    value = rays_per_step * a.steps / dt

    if rank == 0:
        peak = PEAK_BF16 if a.dtype == "bf16" else PEAK_F32
        # dominant kernel = the kernel symbol with the largest event-measured time; the next three are reported beside it ("rooflines")
        # HBM traffic of the dominant kernels (`traffic`): filled in below -- measured in this run when the box has rocprofv3 (default single-GPU fg run,
        # once the training loop has given its memory back), else, declared in traffic_source, the newest committed PMC summary
        pmc = {}
        def roof(name, launches, ms, flops, nbytes):
            # the roofline that binds a kernel = the larger of its two time floors (HBM bytes at 8 TB/s, FLOPs at the dense bf16/fp32
            # MFMA peak).  Training-mode chain kernels write every activation / dZ once: 48.6 GB against 9.6 TFLOP per 8.4 M-sample
            # launch of the basefield backward, so they sit under the HBM roof as well.
            hbm_bound = nbytes / 8.0e12 >= flops / peak
            traffic = pmc.get(name, {}).get("hbm_bytes_per_launch")
            ach_b = nbytes / (ms * 1e-3) / 1e9
            ach_f = flops / (ms * 1e-3) / 1e12
            if hbm_bound:
                r = {"bound": "hbm", "kernel": name, "achieved": round(ach_b, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(ach_b / 8000.0, 4), "traffic": traffic, "mfma_tflops": round(ach_f, 1),
                     # the same launches against the OTHER roof (SURVEY 8d designates the MFMA peak for the fused minimum-traffic design): GEMM FLOPs / time / dense peak
                     "mfma_frac_of_peak": round(ach_f * 1e12 / peak, 4)}
            else:
                r = {"bound": "mfma", "kernel": name, "achieved": round(ach_f, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                     "frac": round(ach_f * 1e12 / peak, 4), "traffic": traffic, "hbm_gbs": round(ach_b, 1)}
            r.update({"launches": launches, "avg_ms": round(ms / launches, 4), "algorithmic_per_launch": {"flop": flops / launches, "bytes": nbytes / launches}})
            return r

        ranked = sorted(((k, v) for k, v in prof.items() if "@" not in k and v[3] > 0), key=lambda kv: -kv[1][1])
        roofline = None
        if ranked:
            name, (launches, ms, flops, nbytes) = ranked[0]
            roofline = roof(name, launches, ms, flops, nbytes)
            roofline.update({"traffic_source": None, "measured": prof_src,
                             "others": [roof(k, *v) for k, v in ranked[1:4]],
                             "kernels_ms_per_step": {k: round(v[1] / n_prof_chunks * len(inputs), 2) for k, v in sorted(prof.items())}})
        flop_per_ray = comp_flop_per_ray(spp // 2) if comp else (multi_flop_per_ray(spp) if multi else spp * FLOP_PER_SAMPLE)
        out = {
            "metric": "rendered rays/sec (fwd+bwd) at 512\u00b2 \u00d7 128 samples; PSNR vs ref", "value": round(value, 1), "unit": "rays/s",
            "n_gpus": dist.get_world_size() if use_dist else 1, "rccl_ranks": dist.get_world_size() if use_dist else None, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": ("human-48-shaped fg+bg composite (MultiFields comp: fg Deformable comp_skel-human_dense, 18 bones + dense post-warp; bg NeRF), "
                                    "%dx%d frame pair, %d + %d samples/ray composed, training graph fwd+bwd+AdamW" % (res, res, spp // 2, spp // 2)) if comp else
                                   ("10-video category model (RAC): fg Deformable comp_skel-quad_dense with num_inst=10 (25 bones + dense post-warp, per-instance codes), "
                                    "%dx%d frame pair of one video, %d samples/ray, training graph fwd+bwd+AdamW" % (res, res, spp)) if multi else
                                   "cat-pikachu fg NeRF (Deformable skel-quad, 25 bones), %dx%d frame pair, %d samples/ray, training graph fwd+bwd+AdamW"
                                   % (res, res, spp), "rays_per_step": rays_per_step, "chunk_rays": 2 * a.chunk_rows * res,
                       "parallelism": "rows dealt round-robin to %d rank(s) and to each rank's chunks, one RCCL all-reduce of the flat fp32 gradient (%d elements) per step" % (world, opt.n),
                       "launch": launch_mode,
                       "chain_kernels": ("weights-stationary (csrc/mlp_kernels_ws.hpp) for the 256-wide nets, wave-resident for the others"
                                         if (a.dtype == "bf16" and mlp.ws_active(mlp.NET_FG_BASE, mlp.PREC_BF16)) else "wave-resident (csrc/mlp_kernels.hpp)"),
                       "optimizer": "lab4d_amd.optim.FlatAdamW: clip_grad_norm_(5.0) + AdamW in 3 launches over one flat buffer; "
                                    "weight gradients accumulated into it by the wgrad kernels; a step whose pre-clip norm exceeds 5 (or is not finite) is "
                                    "discarded on the device like Trainer.check_grad does (steps_discarded below)"},
            "rank_ms_per_step": [round(x, 2) for x in rank_ms], "allreduce_ms_per_step": None if allreduce_ms is None else round(allreduce_ms, 3),
            "replicas_identical": replicas_identical,
            "shared_gpu": ("FUNCTIONAL run: %d ranks on %d GPU(s) over gloo -- value / ms_per_step are NOT a performance measurement" % (world, torch.cuda.device_count())) if a.share_gpu else None,
            "rank_plan": {k: plan[k] for k in ("rows", "chunk_sizes", "rays_per_step", "est_peak_hbm_gib")},
            "peak_hbm_gib": round(peak_hbm / 2**30, 1),
            "whole_graph_tflops": round(value * flop_per_ray / 1e12, 2),
            "whole_graph_frac_of_peak": round(value * flop_per_ray / peak, 4),
            "roofline": roofline,
            # sanity of the timed work: the loss of the last chunk and whether every parameter is still finite after the timed optimizer steps
            "emulated": None if not a.emulate_rank_of else {"rank_0_of": a.emulate_rank_of, "note": "rank 0's share of the strong-scaling job on one GPU, no collective: "
                         "ms_per_step is the per-rank time an %d-GPU run is bounded by (plus its all-reduce of %.1f MB)" % (a.emulate_rank_of, opt.n * 4 / 1e6)},
            "steps_discarded_by_check_grad": n_skipped,
            "sustained": sustained,
            "loss_last_chunk": loss_last, "params_finite": bool(all(bool(torch.isfinite(p).all()) for p in params)),
        }
        if eval_result is not None:
            out["eval_forward_only"] = eval_result
        # the headline is complete here: leave a copy on stderr (and under /tmp) BEFORE the extras run, so that a run cut off inside them
        # (driver timeout) has not lost the measurement.  stdout still carries exactly one line, written at the very end.
        EXTRAS.start()  # the budget covers what follows, not the headline run above
        try:
            head = json.dumps({k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "loss_last_chunk", "params_finite")})
            print("[bench headline, extras still to run] " + head, file=sys.stderr, flush=True)
            with open("/tmp/lab4d_bench_headline.json", "w") as fh:
                fh.write(json.dumps(out, default=str) + "\n")
        except Exception:
            pass
        if comp:
            out["metric"] = "rendered rays/sec (fwd+bwd), fg+bg composite at 512\u00b2 (BASELINE configs[2] per-GPU shape; not the headline metric)"
        if multi:
            out["metric"] = "rendered rays/sec (fwd+bwd), 10-instance category model at 512\u00b2 (BASELINE configs[3] per-GPU shape; not the headline metric)"
        if world == 1 and not comp and not multi and not a.no_extras:
            try:
                out["psnr_vs_ref_db"] = psnr_vs_reference(dev)
            except Exception as e:
                out["psnr_vs_ref_db"] = {"error": repr(e)[:200]}
        if world == 1 and a.dtype == "bf16" and a.config == "fg" and not (a.no_extras or a.no_fp32_leg or a.emulate_rank_of or a.trace):
            # the path that carries the 1e-4 parity claim, at the same shape, in the same line (the reference computes in fp32 only)
            loop = opt = params = inputs = step = None
            import gc
            gc.collect()
            mlp.clear_caches()
            torch.cuda.empty_cache()
            pmc, pmc_why = pmc_traffic_in_run()
            out["fp32_leg"] = fp32_leg(a)
            out["other_configs"] = other_configs_legs()
        else:
            pmc, pmc_why = None, "not the default single-GPU fg run"
        if roofline is not None:
            if pmc is None:
                pmc = {}
                for name_ in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
                    pmc_path = os.path.join(ROOT, "profiles", name_)
                    if os.path.exists(pmc_path):
                        pmc = json.load(open(pmc_path))
                        pmc["_source"] = "NOT measured in this run (%s): committed profiles/%s -- %s" % (pmc_why, name_, pmc.get("_source"))
                        break
            for r_ in [roofline] + roofline.get("others", []):
                r_["traffic"] = pmc.get(r_["kernel"], {}).get("hbm_bytes_per_launch")
            roofline["traffic_source"] = pmc.get("_source") if roofline["traffic"] else None
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(res, spp, a.cpu_rays, a.config)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()


if __name__ == "__main__":
    main()
