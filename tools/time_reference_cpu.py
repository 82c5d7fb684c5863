"""Time the REFERENCE's own code (lab4d-org/lab4d under /root/reference, imported through oracle/ref_shim.py) on this container's
CPU cores, per SURVEY 8d: identical synthetic weights / rays as the GPU bench, training mode, forward + backward of
Deformable.query_field -> render_pixel -> dvr_model losses; 1 warm-up + median of 3 timed passes.
  C1: 64x64 crop of a frame pair, 64 samples/ray (8,192 rays)            -- BASELINE configs[0]
  C2: 512x512, 128 samples/ray: 4 chunks of 8,192 rays (a fixed subset, rays are independent), scaled linearly -- configs[1]
Writes profiles/r02_cpu_reference.json.  Build-container only (the GPU box has no reference tree); bench.py cites the file."""
import json
import os
import sys
import time
from functools import partial

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import ref_shim  # noqa: E402
from lab4d_amd import synthetic  # noqa: E402
import make_golden as MG  # noqa: E402

threads = os.cpu_count()
torch.set_num_threads(threads)
ns = ref_shim.load()
import importlib  # noqa: E402
model = importlib.import_module("lab4d.engine.model").dvr_model
from oracle.lab4d_oracle import DEFAULT_LOSS_WT  # noqa: E402


def one_case(res, D, rays_per_frame_chunks):
    P = synthetic.make_weights(0)
    f = MG.build_reference_field(ns, P)
    f.train()
    fr = MG.frames_from_reference(f, synthetic.make_frames(1, 2, res))
    ns.nerf.sample_cam_rays = partial(ns.render_utils.sample_cam_rays, n_depth=D)
    full = synthetic.make_rays(res, 2)

    def one_pass(hxy, seed):
        N = hxy.shape[1]
        batch = synthetic.make_targets(seed, 2, N, res, hxy)
        sd = MG.samples_dict_of(fr, hxy, batch["feature"])
        feat_dict, deltas, aux = f.query_field(sd, flow_thresh=float(res))
        rendered = ns.render_utils.render_pixel(feat_dict, deltas)
        aux_fg = dict(aux)
        aux_fg.update(rendered)
        results = {"rendered": dict(rendered), "aux_dict": {"fg": aux_fg}}
        config = {"field_type": "fg", "train_res": res}
        L = {}
        model.compute_recon_loss(L, results, batch, config)
        model.mask_losses(L, batch, config)
        L["reg_eikonal"], L["reg_deform_cyc"] = rendered["eikonal"], aux_fg["cyc_dist"]
        L["reg_delta_skin"], L["reg_skin_entropy"] = aux_fg["delta_skin"], aux_fg["skin_entropy"]
        config.update(DEFAULT_LOSS_WT)
        model.apply_loss_weights(L, config)
        f.zero_grad()
        sum(L.values()).backward()

    rates = []
    for ci, (a, b) in enumerate(rays_per_frame_chunks):
        hxy = full[:, a:b].contiguous()
        one_pass(hxy, 100 + ci)  # warm-up
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            one_pass(hxy, 100 + ci)
            ts.append(time.perf_counter() - t0)
        rates.append(2 * (b - a) / sorted(ts)[1])
        print("res", res, "D", D, "chunk", ci, "rays", 2 * (b - a), "median s", round(sorted(ts)[1], 2), "rays/s", round(rates[-1], 1), flush=True)
    return rates


out = {"what": "the reference's own Deformable('skel-quad').query_field + render_pixel + dvr_model losses, forward + backward, training mode, "
               "CPU PyTorch in the build container (oracle/ref_shim.py import shims, no source edits); 1 warm-up + median of 3 passes",
       "cores": threads, "torch": torch.__version__}
r1 = one_case(64, 64, [(0, 4096)])
out["C1_64x64_crop_pair_64spp"] = {"rays_per_pass": 8192, "rays_per_s": round(r1[0], 1)}
n = 512 * 512
r2 = one_case(512, 128, [(i * n // 4 + n // 8, i * n // 4 + n // 8 + 4096) for i in range(4)])
out["C2_512x512_128spp_subset"] = {"chunks": 4, "rays_per_chunk": 8192, "rays_per_s_per_chunk": [round(x, 1) for x in r2],
                                   "rays_per_s": round(sum(r2) / len(r2), 1),
                                   "note": "4 x 8,192-ray subset of the 524,288 rays of a frame pair; rays are independent, the rate scales linearly"}
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_cpu_reference.json"), "w"), indent=1)
print(json.dumps(out))
