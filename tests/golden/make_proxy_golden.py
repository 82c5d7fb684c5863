"""Golden vectors for the proxy-geometry refresh from the REFERENCE's own code: the dense grid query inside
`marching_cubes` (geom_utils.py:445-476: sample_grid -> sdf_func / visibility_func, as `NeRF.extract_canonical_mesh` wires them,
nerf.py:303-343), `NeRF.update_aabb` and `NeRF.update_near_far` (nerf.py:345-376).  Run in the build container only:
    python tests/golden/make_proxy_golden.py        -> tests/golden/proxy.pt"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
OUT_DIR = os.environ.get("LAB4D_GOLDEN_OUT", HERE)
from oracle import ref_shim  # noqa: E402
from lab4d_amd import synthetic  # noqa: E402
import make_golden as MG  # noqa: E402


def main():
    ns = ref_shim.load()
    seed, G = 71, 12
    P = synthetic.make_weights(seed, sdf_bias=-0.02)
    f = MG.build_reference_field(ns, P)
    f.eval()
    out = {"meta": {"seed": seed, "grid_size": G, "sdf_bias": -0.02}}
    with torch.no_grad():
        # the volume marching_cubes meshes, exactly as extract_canonical_mesh sets it up (inst_id=None: mean instance)
        box = ns.geom_utils.extend_aabb(f.aabb, factor=0.5)
        grid = ns.geom_utils.sample_grid(box, G)
        sdf = ns.geom_utils.eval_func_chunk(lambda xyz: f.forward(xyz, inst_id=None, get_density=False), grid, chunk_size=500)
        vis = ns.geom_utils.eval_func_chunk(lambda xyz: f.vis_mlp(xyz, inst_id=None) > 0, grid, chunk_size=500)
        out.update({"aabb": f.aabb.clone(), "box": box.clone(), "grid": grid.clone(), "sdf": sdf.reshape(G, G, G).clone(), "vis": vis.reshape(G, G, G).clone(),
                    "sdf_min_max": (float(sdf.min()), float(sdf.max()))})
        # the same grid with the annealing window live (MultiFields.set_alpha ramps pos_embedding.alpha over the first steps;
        # extract_canonical_mesh calls self.forward, which applies it): alpha = 0.7 of 10 octaves
        f.pos_embedding.set_alpha(0.7)
        f.pos_embedding_color.set_alpha(0.7)
        sdf_a = ns.geom_utils.eval_func_chunk(lambda xyz: f.forward(xyz, inst_id=None, get_density=False), grid, chunk_size=500)
        f.pos_embedding.set_alpha(None)
        f.pos_embedding_color.set_alpha(None)
        out.update({"alpha": 0.7, "sdf_alpha": sdf_a.reshape(G, G, G).clone()})
        # a stand-in proxy mesh (marching cubes itself needs skimage, absent here): a jittered sphere's vertices and bounds
        g = torch.Generator().manual_seed(seed)
        verts = torch.nn.functional.normalize(torch.randn(200, 3, generator=g), dim=-1) * (0.1 + 0.02 * torch.rand(200, 1, generator=g))

        class Mesh:
            vertices = verts.numpy().astype(np.float64)
            bounds = np.stack([verts.numpy().min(0), verts.numpy().max(0)], 0).astype(np.float64)

        f.proxy_geometry = Mesh()
        aabb0 = f.aabb.clone()
        f.update_aabb(beta=0.9)
        out.update({"verts": verts, "aabb_before": aabb0, "aabb_after": f.aabb.clone()})
        nf0 = torch.rand(f.near_far.shape, generator=g) + 0.2
        f.near_far.data.copy_(nf0)
        # the camera MLP of a freshly constructed field has a zero base rotation until mlp_init fits it to the dataset's cameras
        # (nnutils/pose.py:86-114); give it a proper one and a translation in front of the camera so that near / far are meaningful
        f.camera_mlp.base_quat.data.copy_(torch.nn.functional.normalize(torch.randn(f.camera_mlp.base_quat.shape, generator=g), dim=-1))
        f.camera_mlp.trans[-1].bias.data.copy_(torch.tensor([0.01, -0.02, 0.6]))
        quat, trans = f.camera_mlp.get_vals()
        f.update_near_far(beta=0.9)
        out.update({"near_far_before": nf0, "near_far_after": f.near_far.data.clone(), "cam_quat": quat.clone(), "cam_trans": trans.clone(),
                    "frame_mapping": f.camera_mlp.time_embedding.frame_mapping.clone(),
                    "get_near_far": ns.geom_utils.get_near_far(verts, ns.quat_transform.quaternion_translation_to_se3(quat, trans)).clone()})
    path = os.path.join(OUT_DIR, "proxy.pt")
    torch.save(out, path)
    print("proxy ->", path, os.path.getsize(path) // 1024, "KiB", "sdf range", out["sdf_min_max"], "visible", float(out["vis"].float().mean()))


if __name__ == "__main__":
    main()
