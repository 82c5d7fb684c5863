"""Import shim for the *real* reference (lab4d-org/lab4d under /root/reference).

TEST INFRASTRUCTURE ONLY.  Used by tests/golden/make_golden.py (in the build
container, where /root/reference exists) to run the reference's own Python on
CPU and dump golden vectors.  Nothing in the product path, the -m gpu tests,
smoke() or bench.py imports this file: /root/reference does not exist on the
GPU box.

Why a shim is needed:
  * lab4d/__init__.py imports every module (trimesh, cv2, absl ... missing here)
    and JIT-builds a CUDA extension            -> bypass the package __init__
  * quat_transform.py:15-16 imports the native `quaternion` package   -> stub
  * quat_transform.py:106-113 CPU `_quaternion_mul` rejects 3-vectors, only the
    CUDA kernel (quaternion.cu:46-57) zero-pads them        -> pad on CPU
  * nerf.py:491 hard-codes device="cuda"                      -> patched call
No file under /root/reference is modified.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = os.environ.get("LAB4D_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "lab4d", "nnutils"))


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _Stub(self.__name__ + "." + name)
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        raise RuntimeError("stubbed third-party call: %s" % self.__name__)


class _Sphere:
    """Minimal stand-in for trimesh.creation.uv_sphere (deformable.py:86-93)."""

    def __init__(self, radius, count):
        nu, nv = count
        th = np.linspace(0, np.pi, nu + 1)
        ph = np.linspace(0, 2 * np.pi, nv + 1)[:-1]
        v = [[0, 0, radius]]
        for t in th[1:-1]:
            for p in ph:
                v.append([radius * np.sin(t) * np.cos(p), radius * np.sin(t) * np.sin(p), radius * np.cos(t)])
        v.append([0, 0, -radius])
        self.vertices = np.asarray(v, dtype=np.float64)
        self.faces = np.zeros((0, 3), dtype=np.int64)

    @property
    def bounds(self):
        return np.stack([self.vertices.min(0), self.vertices.max(0)], 0)


def _corners(bounds):
    b = np.asarray(bounds)
    out = []
    for i in range(8):
        out.append([b[(i >> 0) & 1, 0], b[(i >> 1) & 1, 1], b[(i >> 2) & 1, 2]])
    return np.asarray(out)


_loaded = {}


def load():
    """Returns a namespace of reference modules, importing them once."""
    if _loaded:
        return _loaded["ns"]
    if not available():
        raise RuntimeError("reference tree not found at %s" % REF_ROOT)

    # 1. third-party stubs
    # (tensorboard / matplotlib: only lab4d.engine.trainer and its vis_utils want them -- imported by the binding tests of lab4d_amd.patch)
    for name in ["trimesh", "pysdf", "cv2", "skimage", "imageio", "absl", "absl.app", "absl.flags", "tqdm", "tensorboard", "torch.utils.tensorboard",
                 "matplotlib", "matplotlib.pyplot", "matplotlib.cm", "einops"]:
        if name in sys.modules:
            continue
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = _Stub(name)
    tm = sys.modules["trimesh"]
    if isinstance(tm, _Stub):
        tm.creation = _Stub("trimesh.creation")
        tm.creation.uv_sphere = lambda radius=1.0, count=(4, 4): _Sphere(radius, count)
        tm.bounds = _Stub("trimesh.bounds")
        tm.bounds.corners = _corners
        tm.Trimesh = lambda *a, **k: _Sphere(0.0, (2, 2))
        # NeRF.init_proxy (nerf.py:247) loads the bg proxy mesh from disk: a unit sphere stands in (only aabb / near-far use it)
        tm.load = lambda *a, **k: _Sphere(1.0, (4, 4))
    sk = sys.modules["skimage"]
    if isinstance(sk, _Stub):
        sys.modules["skimage.measure"] = sk.measure

    # 2. native quaternion package stub (filled in after quat_transform import)
    q = types.ModuleType("quaternion")
    q.quaternion_conjugate = lambda x: torch.cat((x[..., :1], -x[..., 1:]), -1)
    q.quaternion_mul = None
    q.mat3x3_inv = lambda m: torch.linalg.inv(m)
    sys.modules["quaternion"] = q

    # 3. bypass lab4d/__init__.py
    for pkg in ["lab4d", "lab4d.utils", "lab4d.nnutils", "lab4d.engine", "lab4d.third_party", "lab4d.dataloader"]:
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF_ROOT, *pkg.split("."))]
        sys.modules[pkg] = m

    qt = importlib.import_module("lab4d.utils.quat_transform")
    _mul4 = qt._quaternion_mul

    def _mul_pad(a, b):
        # mirrors quaternion.cu:46-57: a 3-vector operand is a pure quaternion (w=0)
        if a.shape[-1] == 3:
            a = torch.cat([torch.zeros_like(a[..., :1]), a], -1)
        if b.shape[-1] == 3:
            b = torch.cat([torch.zeros_like(b[..., :1]), b], -1)
        a, b = torch.broadcast_tensors(a, b)
        return _mul4(a, b)

    qt._quaternion_mul = _mul_pad
    q.quaternion_mul = _mul_pad

    ns = types.SimpleNamespace()
    ns.quat_transform = qt
    for short, full in [
        ("render_utils", "lab4d.utils.render_utils"),
        ("geom_utils", "lab4d.utils.geom_utils"),
        ("transforms", "lab4d.utils.transforms"),
        ("loss_utils", "lab4d.utils.loss_utils"),
        ("torch_utils", "lab4d.utils.torch_utils"),
        ("embedding", "lab4d.nnutils.embedding"),
        ("base", "lab4d.nnutils.base"),
        ("nerf", "lab4d.nnutils.nerf"),
        ("feature", "lab4d.nnutils.feature"),
        ("deformable", "lab4d.nnutils.deformable"),
        ("warping", "lab4d.nnutils.warping"),
        ("skinning", "lab4d.nnutils.skinning"),
        ("visibility", "lab4d.nnutils.visibility"),
        ("multifields", "lab4d.nnutils.multifields"),
    ]:
        setattr(ns, short, importlib.import_module(full))

    # 4. nerf.py:491 device="cuda"
    _orig_tensor = torch.tensor

    def _tensor(*a, **k):
        if k.get("device", None) == "cuda" and not torch.cuda.is_available():
            k["device"] = "cpu"
        return _orig_tensor(*a, **k)

    ns.nerf.torch = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
    ns.nerf.torch.tensor = _tensor
    _loaded["ns"] = ns
    return ns


def synthetic_data_info(T=64):
    """T frames, camera orbiting on a circle at z=3."""
    rt = np.zeros((T, 4, 4), dtype=np.float32)
    for t in range(T):
        a = 2 * np.pi * t / T
        rt[t] = np.array(
            [[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 3.0], [0, 0, 0, 1]],
            dtype=np.float32,
        )
    return {
        "rtmat": rt,
        "geom_path": "",
        "frame_info": {
            "frame_offset": np.asarray([0, T]),
            "frame_offset_raw": np.asarray([0, T]),
            "frame_mapping": list(range(T)),
        },
    }
