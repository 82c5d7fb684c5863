"""Proxy-geometry refresh on the device: the per-sample work of `NeRF.extract_canonical_mesh` -- the dense grid
query of the signed-distance field and the visibility field that feeds marching cubes (nnutils/nerf.py:303-343,
utils/geom_utils.py:392-476) -- and the bound updates that follow it, `NeRF.update_aabb` / `update_near_far`
(nerf.py:345-376, geom_utils.py:344-362).

The grid query is the chain kernels' inference mode (nothing stored, MFMA-bound): one "frame" of grid_size^3 samples under one
instance code, sdf head only (no density, no colour), then the visibility net.  Marching cubes itself is the reference's CPU
`skimage.measure.marching_cubes` on the returned volume (out of the hot path); what comes back from it -- the vertices / bounds
of the new proxy mesh -- goes into `update_aabb` / `update_near_far` here, which keep `aabb` and `near_far` on the device."""
import torch

from . import deformable as DF
from . import mlp
from . import quat_utils as Q


def sample_grid(aabb, grid_size):
    """geom_utils.sample_grid (geom_utils.py:392-406): (grid_size^3, 3) points, x slowest (torch.cartesian_prod order)."""
    ax = [torch.linspace(float(aabb[0][i]), float(aabb[1][i]), grid_size, device=aabb.device) for i in range(3)]
    return torch.cartesian_prod(*ax)


@torch.no_grad()
def grid_query(P, aabb, grid_size=64, code_base=None, code_vis=None, prec=mlp.PREC_BF16, use_visibility=True, extend=0.5, chunk=1 << 22, alpha=None,
               kind="fg"):
    """The volume `marching_cubes` meshes (geom_utils.py:445-476): sdf (G,G,G) fp32 and visibility mask (G,G,G) bool over
    extend_aabb(aabb, extend) (extract_canonical_mesh's `use_extend_aabb`, nerf.py:333-336).  code_base / code_vis: (1,32) instance
    codes of the basefield / visibility CondMLPs -- `inst_embedding(inst_id)` or the mean embedding for inst_id=None
    (base.py:130-134); default: the mean of P's embedding tables.  alpha: the live annealing state `pos_embedding.alpha` of
    the field (embedding.py:112-125; MultiFields.set_alpha, multifields.py:108-116, ramps it over the first steps, and
    extract_canonical_mesh calls self.forward, which applies it; the visibility field's own embedding is never annealed); None = no window.
    kind: "fg" (NeRF W=256, D=8, 10 frequencies: LAB4D_NET_FG_BASE) or "bg" (multifields.py:86-93: W=128, D=5, 6 frequencies: LAB4D_NET_BG_BASE) --
    MultiFields.update_geometry_aux (trainer.py:247) and MultiFields.extract_canonical_meshes mesh EVERY field, the background included; both
    kinds carry the same VisField (95 -> 64 -> 64 -> 1).  Returns (sdf, vis, grid_aabb)."""
    net = {"fg": mlp.NET_FG_BASE, "bg": mlp.NET_BG_BASE}[kind]
    box = DF.extend_aabb(aabb, extend) if extend else aabb
    pts = sample_grid(box, grid_size)
    if code_base is None:
        code_base = P["basefield.inst_embedding.mapping.weight"].mean(0, keepdim=True)
    if code_vis is None:
        code_vis = P["vis_mlp.basefield.inst_embedding.mapping.weight"].mean(0, keepdim=True)
    sdf, vis = [], []
    fw = DF.posenc_window(alpha, mlp.describe(net).n_freq, pts.device)
    for i in range(0, pts.shape[0], chunk):  # eval_func_chunk (geom_utils.py:425-440); one chunk up to 128^3
        x = pts[i:i + chunk].contiguous()
        sdf.append(mlp.run_chain(net, prec, P, x, x.shape[0], conds={0: code_base, 4: code_base}, freq_w=fw))
        if use_visibility:
            vis.append(mlp.run_chain(mlp.NET_VIS, prec, P, x, x.shape[0], conds={0: code_vis}) > 0)
    G = grid_size
    sdf = torch.cat(sdf, 0).view(G, G, G)
    vis = torch.cat(vis, 0).view(G, G, G) if use_visibility else torch.ones(G, G, G, dtype=torch.bool, device=sdf.device)
    return sdf, vis, box


def grid_to_world(verts01, box):
    """marching_cubes' vertex transform from the unit cube to the box (geom_utils.py:492-493)."""
    return verts01 * (box[1:] - box[:1]) + box[:1]


@torch.no_grad()
def update_aabb(aabb, bounds, beta=0.9):
    """NeRF.update_aabb (nerf.py:345-356): EMA of the field's aabb towards the proxy mesh's bounds (2,3)."""
    return aabb * beta + bounds.to(aabb) * (1 - beta)


@torch.no_grad()
def get_near_far(pts, quat, trans, tol_fac=1.5):
    """geom_utils.get_near_far (geom_utils.py:344-362) with the object-to-camera transforms as (quat (M,4), trans (M,3)) -- the
    form CameraMLP.get_vals returns (the reference converts them to 4x4 matrices first, nerf.py:366-367): depth range of the proxy
    vertices in every camera, widened by (tol_fac - 1) of its extent, clamped at 1e-3."""
    z = DF.rigid_apply(quat, trans, pts[None].expand(quat.shape[0], -1, -1))[..., 2]
    pmax, pmin = z.max(-1)[0], z.min(-1)[0]
    delta = (pmax - pmin) * (tol_fac - 1)
    return torch.stack([pmin - delta, pmax + delta], -1).clamp(min=1e-3)


@torch.no_grad()
def update_near_far(near_far, frame_mapping, pts, quat, trans, beta=0.9):
    """NeRF.update_near_far (nerf.py:358-376): EMA of the per-frame near / far planes (rows `frame_mapping` of the (T_raw, 2) table)."""
    nf = get_near_far(pts, quat, trans)
    out = near_far.clone()
    out[frame_mapping] = near_far[frame_mapping] * beta + nf * (1 - beta)
    return out
