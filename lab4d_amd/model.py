"""Host-side mirror of the renderer entry points of lab4d/engine/model.py for field_type == "fg":
`render_samples` (model.py:328-361) and `render_samples_chunk` (model.py:259-326, pixel chunking along N with
`chunk_size_n = ceil(chunk_size // M)` and concatenation of every output key).

`samples_dict` has the reference's keys (model.py:236-257, deformable.py:254-289): "Kinv", "field2cam", "frame_id",
"inst_id", "near_far", "hxy", "feature", "t_articulation", "rest_articulation" -- plus the three per-frame codes the
reference computes inside the field from frame_id ("t_embed", "t_embed_mean", "appr_code"); those per-frame modules are
outside the hot path and are handed in by the caller.
"""
import math

import torch

from . import deformable as DF
from . import mlp, synthetic

PER_FRAME_KEYS = ["Kinv", "field2cam", "frame_id", "inst_id", "near_far", "t_articulation", "rest_articulation", "t_embed",
                  "t_embed_mean", "appr_code"]


def _frames(P, samples_dict):
    fr = {k: samples_dict[k] for k in PER_FRAME_KEYS}
    return synthetic.add_codes(fr, P)


def render_samples(P, samples_dict, flow_thresh=None, training=True, rng=None, n_depth=64, alpha=None, prec=mlp.PREC_F32):
    """dvr_model.render_samples: query the field, composite, return {"rendered", "aux_dict"}."""
    fr = _frames(P, samples_dict)
    hxy = samples_dict["hxy"]
    if training:
        fr["feature"] = samples_dict["feature"]
        if rng is None:
            M, N = hxy.shape[:2]
            S = M * N * n_depth
            rng = {"eik_inds": torch.randperm(M * N, device=hxy.device)[: max(M * N // 16, 1)],
                   "match_perm": torch.randperm(S, device=hxy.device)[: min(1024, S)]}
        return DF.render_train(P, fr, hxy, rng, flow_thresh=flow_thresh, n_depth=n_depth, alpha=alpha, prec=prec)
    out = DF.render_eval(P, fr, hxy, n_depth=n_depth, alpha=alpha, prec=prec)
    out.pop("debug", None)
    return out


def render_samples_chunk(P, samples_dict, flow_thresh=None, chunk_size=8192, **kw):
    """dvr_model.render_samples_chunk: split the rays of every frame into chunks of ceil(chunk_size // M) pixels."""
    hxy = samples_dict["hxy"]
    M, N = hxy.shape[:2]
    num_chunks = int(math.ceil(M * N / chunk_size))
    chunk_n = int(math.ceil(chunk_size // M))
    rendered, aux = {}, {}
    for i in range(num_chunks):
        sd = dict(samples_dict)
        sd["hxy"] = hxy[:, i * chunk_n:(i + 1) * chunk_n]
        if "feature" in sd and sd["feature"] is not None:
            sd["feature"] = samples_dict["feature"][:, i * chunk_n:(i + 1) * chunk_n]
        if sd["hxy"].shape[1] == 0:
            continue
        res = render_samples(P, sd, flow_thresh=flow_thresh, **kw)
        for k, v in res["rendered"].items():
            rendered.setdefault(k, []).append(v)
        for k, v in res["aux_dict"]["fg"].items():
            aux.setdefault(k, []).append(v)
    return {"rendered": {k: torch.cat(v, 1) for k, v in rendered.items()},
            "aux_dict": {"fg": {k: torch.cat(v, 1) for k, v in aux.items()}}}
