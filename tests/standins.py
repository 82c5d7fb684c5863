"""Minimal stand-ins for the reference's module objects, for driving lab4d_amd.patch on the GPU box (where /root/reference does
not exist).  TEST INFRASTRUCTURE ONLY.

A stand-in field is a tree of torch.nn.Modules whose `named_parameters()` are exactly the reference's state_dict names (built
from a flat weight dict) and which carries the handful of attributes / per-frame sub-modules the adapters read:
`pos_embedding.alpha / N_freqs`, `dir_embedding.out_channels`, `sdf.in_features`, `appr_channels`, `training`,
`<CondMLP>.inst_embedding(inst_id) / .get_mean_embedding()`, `appr_embedding.get_vals(frame_id)`,
`warp.skinning_model.time_embedding(frame_id) / .get_mean_embedding(device)`, `warp.skinning_model.symm_idx`,
`warp.post_warp.time_embedding(frame_id)`, `warp.articulation.get_vals_and_mean(frame_id)`.
Per-frame modules that are outside the hot path (time / appearance embeddings, articulation) are tables holding the values the
REFERENCE's own modules produced for the fixture's frame ids (stored in the fixture by tests/golden/make_golden.py)."""
import torch
import torch.nn as nn


class Node(nn.Module):
    pass


class InstEmbedding(Node):
    """embedding.py:230-298: `mapping.weight` (num_inst, C); one instance -> row 0 whatever the id."""

    def forward(self, inst_id):
        w = self.mapping.weight
        return w[torch.zeros_like(inst_id) if w.shape[0] == 1 else inst_id]

    def get_mean_embedding(self):
        return self.mapping.weight.mean(0)


class FrameTable:
    """A per-frame module replaced by the rows the reference produced: table[i] belongs to frame_ids[i]."""

    def __init__(self, frame_ids, table, mean=None):
        self.ids, self.table, self.mean = [int(i) for i in frame_ids], table, mean

    def rows(self, frame_id):
        return self.table[torch.tensor([self.ids.index(int(i)) for i in frame_id], device=self.table.device)]

    def __call__(self, frame_id):
        return self.rows(frame_id)

    get_vals = __call__

    def get_mean_embedding(self, device):
        return self.mean.to(device)


class Articulation:
    def __init__(self, frame_ids, t_art, rest_art):
        self.t = (FrameTable(frame_ids, t_art[0]), FrameTable(frame_ids, t_art[1]))
        self.r = (FrameTable(frame_ids, rest_art[0]), FrameTable(frame_ids, rest_art[1]))

    def get_vals_and_mean(self, frame_id):
        return (self.t[0](frame_id), self.t[1](frame_id)), (self.r[0](frame_id), self.r[1](frame_id))


class SkinningWarp(Node):  # the adapters dispatch on the class NAME (patch.warp_kind)
    pass


class ComposedWarp(Node):
    pass


class IdentityWarp(Node):  # fg_motion "rigid" (warping.py:59-91)
    pass


class DenseWarp(Node):  # fg_motion "dense" (warping.py:94-170): forward_map / backward_map CondMLPs + its own time embedding
    pass


class PosEmb:
    def __init__(self, n_freqs, alpha=None):
        self.N_freqs, self.alpha = n_freqs, alpha
        self.out_channels = 0 if n_freqs == -1 else 3 * (2 * n_freqs + 1)


def _tree(root, P):
    for name, v in P.items():
        if not torch.is_tensor(v) or not v.dtype.is_floating_point or name == "aabb":
            continue
        node = root
        parts = name.split(".")
        for part in parts[:-1]:
            if not hasattr(node, part):
                node.add_module(part, InstEmbedding() if part == "inst_embedding" else Node())
            node = getattr(node, part)
        node.register_parameter(parts[-1], v if isinstance(v, nn.Parameter) else nn.Parameter(v))
    return root


def fg_field(P, frames, composed=False, alpha=None, training=True, motion=None):
    """Stand-in for Deformable("skel-quad" | "comp_skel-quad_dense" | "rigid" | "dense") holding the device weights P (flat, reference names)
    and the fixture's per-frame values `frames` (t_embed, t_embed_mean, appr_code, articulations, frame_id[, t_embed_dense])."""
    f = Node()
    warp = {"rigid": IdentityWarp, "dense": DenseWarp}[motion]() if motion in ("rigid", "dense") else (ComposedWarp() if composed else SkinningWarp())
    f.add_module("warp", warp)
    _tree(f, P)
    f.register_buffer("aabb", P["aabb"])
    fid = frames["frame_id"]
    f.pos_embedding, f.pos_embedding_color, f.dir_embedding = PosEmb(10, alpha), PosEmb(12, alpha), PosEmb(-1)
    f.sdf.in_features = f.sdf.weight.shape[1]
    f.appr_channels = 32
    f.appr_embedding = FrameTable(fid, frames["appr_code"])
    if motion in ("rigid", "dense"):
        if motion == "dense":
            warp.time_embedding = FrameTable(fid, frames["t_embed_dense"])
        f.train(training)
        return f
    sk = warp.skinning_model
    sk.symm_idx = [int(i) for i in P["warp.skinning_model.symm_idx"]]
    sk.time_embedding = FrameTable(fid, frames["t_embed"], frames["t_embed_mean"])
    warp.articulation = Articulation(fid, frames["t_articulation"], frames["rest_articulation"])
    if composed:
        warp.post_warp.time_embedding = FrameTable(fid, frames["t_embed_dense"])
    f.train(training)
    return f


def bg_field(P, training=True):
    """Stand-in for the background NeRF(num_freq_xyz=6, num_freq_dir=0, appr_channels=0) (multifields.py:86-93)."""
    f = Node()
    _tree(f, P)
    f.pos_embedding, f.pos_embedding_color, f.dir_embedding = PosEmb(6), PosEmb(8), PosEmb(0)
    f.sdf.in_features = f.sdf.weight.shape[1]
    f.appr_channels = 0
    f.train(training)
    return f
