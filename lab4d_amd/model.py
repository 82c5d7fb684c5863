"""Host-side mirror of the renderer entry points of lab4d/engine/model.py for field_type == "fg" and "comp":
`render_samples` (model.py:328-361) and `render_samples_chunk` (model.py:259-326, pixel chunking along N with
`chunk_size_n = ceil(chunk_size // M)` and concatenation of every output key).

`samples_dict` has the reference's keys (model.py:236-257, deformable.py:254-289): "Kinv", "field2cam", "frame_id",
"inst_id", "near_far", "hxy", "feature", "t_articulation", "rest_articulation" -- plus the three per-frame codes the
reference computes inside the field from frame_id ("t_embed", "t_embed_mean", "appr_code"); those per-frame modules are
outside the hot path and are handed in by the caller.

field_type "comp": `samples_dict` = {"fg": {...as above...}, "bg": {"Kinv", "field2cam", "frame_id", "inst_id", "near_far",
"hxy"}} (one dict per category, multifields.py:300-337) and `P` = {"fg": fg parameters, "bg": bg parameters}.
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


BG_FRAME_KEYS = ["Kinv", "field2cam", "frame_id", "inst_id", "near_far"]


def is_comp(samples_dict):
    return isinstance(samples_dict.get("fg"), dict) and isinstance(samples_dict.get("bg"), dict)


def _render_samples_comp(P, samples_dict, flow_thresh, training, rng, n_depth, alpha, prec):
    """field_type "comp": both fields on the same rays -> compose_fields -> render_pixel (model.py:328-361)."""
    sf, sb = samples_dict["fg"], samples_dict["bg"]
    fr_fg = _frames(P["fg"], sf)
    fr_bg = synthetic.add_bg_codes({k: sb[k] for k in BG_FRAME_KEYS}, P["bg"])
    hxy = sf["hxy"]
    if training:
        fr_fg["feature"] = sf["feature"]
        if rng is None:
            M, N = hxy.shape[:2]
            S = M * N * n_depth
            rng = {"eik_inds": torch.randperm(M * N, device=hxy.device)[: max(M * N // 16, 1)],
                   "eik_inds_bg": torch.randperm(M * N, device=hxy.device)[: max(M * N // 16, 1)],
                   "match_perm": torch.randperm(S, device=hxy.device)[: min(1024, S)]}
        return DF.render_train_comp(P["fg"], fr_fg, P["bg"], fr_bg, hxy, rng, flow_thresh=flow_thresh, n_depth=n_depth, alpha=alpha, prec=prec)
    out = DF.render_eval_comp(P["fg"], fr_fg, P["bg"], fr_bg, hxy, n_depth=n_depth, alpha=alpha, prec=prec)
    return {"rendered": out["rendered"], "aux_dict": out["aux_dict"]}


def render_samples(P, samples_dict, flow_thresh=None, training=True, rng=None, n_depth=64, alpha=None, prec=mlp.PREC_F32):
    """dvr_model.render_samples: query the field(s), composite, return {"rendered", "aux_dict"}."""
    if is_comp(samples_dict):
        return _render_samples_comp(P, samples_dict, flow_thresh, training, rng, n_depth, alpha, prec)
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
    comp = is_comp(samples_dict)
    cats = ["fg", "bg"] if comp else [None]
    hxy = samples_dict["fg"]["hxy"] if comp else samples_dict["hxy"]
    M, N = hxy.shape[:2]
    num_chunks = int(math.ceil(M * N / chunk_size))
    chunk_n = int(math.ceil(chunk_size // M))

    def cut(sd_in, i):
        sd = dict(sd_in)
        sd["hxy"] = sd_in["hxy"][:, i * chunk_n:(i + 1) * chunk_n]
        if sd_in.get("feature") is not None:
            sd["feature"] = sd_in["feature"][:, i * chunk_n:(i + 1) * chunk_n]
        return sd

    rendered, aux = {}, {}
    for i in range(num_chunks):
        sd = {c: cut(samples_dict[c], i) for c in cats} if comp else cut(samples_dict, i)
        if (sd["fg"] if comp else sd)["hxy"].shape[1] == 0:
            continue
        res = render_samples(P, sd, flow_thresh=flow_thresh, **kw)
        for k, v in res["rendered"].items():
            rendered.setdefault(k, []).append(v)
        for c, d in res["aux_dict"].items():
            for k, v in d.items():
                aux.setdefault(c, {}).setdefault(k, []).append(v)
    return {"rendered": {k: torch.cat(v, 1) for k, v in rendered.items()},
            "aux_dict": {c: {k: torch.cat(v, 1) for k, v in d.items()} for c, d in aux.items()}}
