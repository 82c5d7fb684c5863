"""Host side of the fused posenc + MLP chain (include/lab4d_mlp.h, csrc/mlp_kernels.hpp).

Binds the reference's own parameters (nn.Linear weights in reference layout, state_dict names as
in lab4d/nnutils/{nerf,visibility,feature,skinning}.py) to the kernel's layer tables:
  * columns that multiply the positional embedding are permuted into the kernel's slot order,
  * columns that multiply per-frame conditioning (instance / appearance / time codes) are split
    off and turned into a per-frame bias with one tiny device matmul (base.py:139-146 would
    broadcast them to every sample),
  * weights are packed into MFMA A-fragment order by the device pack kernel, cached on the
    parameter version so an optimizer step invalidates them.
`MlpChain` is a torch.autograd.Function whose backward runs the dgrad chain kernel and the
weight-gradient GEMMs and hands gradients back in the reference layout.
"""
import ctypes
import os
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib

MAXL = 12
NET_FG_BASE, NET_FG_COLOR, NET_VIS, NET_FEAT, NET_SKIN, NET_DENSE, NET_BG_BASE, NET_BG_COLOR, NET_SKIN18, NET_HASH_GEO, NET_HASH_COLOR, NET_DENSE6 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
NET_SKIN_A, NET_SKIN18_A = 12, 13  # the delta-skin nets with linear_1 in per-frame affine form (include/lab4d_mlp.h)
PREC_F32, PREC_BF16 = 0, 1
vp, ci = ctypes.c_void_p, ctypes.c_int


class LayerDesc(ctypes.Structure):
    _fields_ = [(n, ci) for n in ("ke", "kin", "mout", "mout_pad", "relu", "pf_bias", "add_ext", "ext_grad")]


class NetDesc(ctypes.Structure):
    _fields_ = [(n, ci) for n in ("n_layers", "emb_kind", "n_freq", "c_in", "emb_slots", "ke", "c_out")] + [("layers", LayerDesc * MAXL)]


class FwdArgs(ctypes.Structure):
    _fields_ = [("net", ci), ("precision", ci), ("S", ci), ("S_pad", ci), ("ld", ci), ("spf", ci), ("x", vp), ("freq_w", vp),
                ("W", vp * MAXL), ("bias", vp * MAXL), ("pf_bias", vp * MAXL), ("act", vp * MAXL), ("mask", vp * MAXL), ("emb", vp),
                ("ext", vp), ("out", vp), ("x2", vp), ("S_dev", vp), ("frame_idx", vp), ("aff", vp)]


class BwdArgs(ctypes.Structure):
    _fields_ = [("net", ci), ("precision", ci), ("S", ci), ("S_pad", ci), ("ld", ci), ("spf", ci), ("WT", vp * MAXL), ("act", vp * MAXL),
                ("mask", vp * MAXL), ("emb", vp), ("ext", vp), ("d_out", vp), ("ext_gin", vp), ("ext_gout", vp), ("dz", vp * MAXL), ("d_x", vp),
                ("d_x2", vp), ("x", vp), ("aff", vp), ("g_aff", vp)]


class BwdFusedArgs(ctypes.Structure):  # lab4d_mlp_bwd_fused_args (include/lab4d_mlp.h)
    _fields_ = [("net", ci), ("precision", ci), ("S", ci), ("spf", ci), ("x", vp), ("freq_w", vp), ("aff", vp), ("W", vp * MAXL), ("WT", vp * MAXL),
                ("bias", vp * MAXL), ("pf_bias", vp * MAXL), ("d_out", vp), ("d_x", vp), ("g_aff", vp), ("dW", vp * MAXL), ("db", vp * MAXL),
                ("pf_db", vp * MAXL)]


_lib.register("lab4d_mlp_describe", [ci, ctypes.POINTER(NetDesc)])
_lib.register("lab4d_mlp_backward_fused", [ctypes.POINTER(BwdFusedArgs), vp])
_lib.register("lab4d_mlp_fused_backward_supported", [ci, ci, ci])
_lib.register("lab4d_mlp_pack", [ci, ci, ci, ci, vp, ci, vp, vp, vp])
_lib.register("lab4d_mlp_forward", [ctypes.POINTER(FwdArgs), vp])
_lib.register("lab4d_mlp_backward", [ctypes.POINTER(BwdArgs), vp])
_lib.register("lab4d_mlp_forward_tangent", [ctypes.POINTER(FwdArgs), vp])
_lib.register("lab4d_eikonal_tangent_input", [vp, vp, vp, vp, ci, ci, ci, vp, vp])
_lib.register("lab4d_mlp_wgrad", [ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, ci, vp])
_lib.register("lab4d_mlp_wgrad_mapped", [ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp, vp, ci, vp])
_lib.SIGNATURES["lab4d_mlp_packed_bytes"] = [ci, ci, ci]

NET_NAMES = {0: "fg_base", 1: "fg_color", 2: "vis", 3: "feat", 4: "skin", 5: "dense", 6: "bg_base", 7: "bg_color", 8: "skin18", 9: "hash_geo", 10: "hash_color", 11: "dense6",
             12: "skin_a", 13: "skin18_a"}
# algorithmic MACs per sample (real layer shapes incl. conditioning columns)
NET_MACS = {0: 572928 + 256, 1: 158464 + 37248, 2: 10240, 3: 77568, 4: 20736, 5: 39 * 256 + 256 * 256 + 256 * 3,
            6: 100096 + 128, 7: 43392 + 8576, 8: (54 + 160) * 64 + 64 * 64 + 64 * 18, 9: 32 * 64 + 64 * 16, 10: 19 * 64 + 64 * 64 + 64 * 3,
            11: 199 * 256 + 3 * 256 * 256 + 455 * 256 + 256 * 256 + 256 * 3,
            12: 4 * 64 + 64 * 64 + 64 * 25, 13: 4 * 64 + 64 * 64 + 64 * 18}  # affine form: what the kernels execute per sample



KERNEL_NET = {0: "FgBase", 1: "FgColor", 2: "Vis", 3: "Feat", 4: "Skin", 5: "Dense", 6: "BgBase", 7: "BgColor", 8: "Skin18", 9: "HashGeo", 10: "HashColor", 11: "Dense6", 12: "SkinA", 13: "Skin18A"}  # template argument names in csrc/mlp_nets.hpp


WS_NETS = (0, 1, 5, 11)  # fg_base, fg_color, dense, dense6: the 256-wide posenc nets (csrc/mlp_kernels_ws.hpp ws_ok<Net>())


def ws_active(net, prec, dx_only=False):
    """Whether lab4d_mlp_forward / _backward launch the weights-stationary chain kernels for this call (mirrors launch_ws_fwd / launch_ws_bwd in
    csrc/mlp_kernels.hpp: bf16, the 256-wide posenc nets -- training, inference and point-gradient-only modes -- LAB4D_WS unset or non-zero)."""
    e = os.environ.get("LAB4D_WS")
    try:
        on = e is None or int(e.strip() or "0") != 0
    except ValueError:
        on = False  # atoi() of a non-number is 0
    return bool(on and prec == PREC_BF16 and net in WS_NETS)


def chain_kernel_name(kind, net, prec, dx_only=False):
    """Kernel symbol (as the profiles name it) behind a chain launch: k_mlp_fwd<Net> / k_mlp_fwd_ws<Net> / ..."""
    return "k_mlp_%s%s<%s>" % (kind, "_ws" if ws_active(net, prec, dx_only) else "", KERNEL_NET[net])


def wgrad_kernel_name(L, prec):
    """Which kernel lab4d_mlp_wgrad dispatches to for this layer (mirrors the dispatch in csrc/mlp.hip)."""
    if prec == PREC_BF16 and L.mout_pad == 32 and os.environ.get("LAB4D_WGRAD_HEAD_DMA", "1") not in ("", "0") and L.ke + L.kin <= 256:
        return "k_mlp_wgrad_dma<1,%d>" % (2 if L.ke + L.kin <= 128 else 4)  # the <= 32-row heads on the DMA ring (csrc/mlp.hip; LAB4D_WGRAD_HEAD_DMA=0: the pre-DMA kernel)
    if prec == PREC_BF16 and L.mout_pad in (64, 128, 256):
        K = L.ke + L.kin
        nbw = 1 if K <= 64 else (2 if K <= 128 else (3 if K <= 192 else (4 if K <= 256 else (5 if (L.mout_pad == 256 and K <= 320) else 3))))
        return "k_mlp_wgrad_dma<%d,%d>" % (L.mout_pad // 32, nbw)
    if L.mout_pad >= 256:
        return "k_mlp_wgrad_big"
    return "k_mlp_wgrad<%d>" % (4 if L.mout_pad >= 128 else (2 if L.mout_pad >= 64 else 1))


def wgrad_work(L, S_pad, prec):
    """(algorithmic FLOPs, algorithmic HBM bytes) of one wgrad launch: 2*mout*K MAC-flops per sample; both operands
    ([mout_pad + K] feature rows of S_pad samples) are read once, the (mout_pad, K) fp32 result is written once."""
    K = L.ke + L.kin
    esize = 2 if prec == PREC_BF16 else 4
    return (2.0 * S_pad * L.mout * K, float((L.mout_pad + K) * S_pad * esize + L.mout_pad * K * 4))


_DESC = {}


def describe(net):
    if net not in _DESC:
        d = NetDesc()
        _lib.check(_lib.lib().lab4d_mlp_describe(net, ctypes.byref(d)), "mlp_describe")
        _DESC[net] = d
    return _DESC[net]


def store_dtype(prec):
    return torch.bfloat16 if prec == PREC_BF16 else torch.float32


def posenc_slot_to_ref_channel(n_freq, ke):
    """Kernel slot order -> reference channel order (embedding.py:96-108: [x, (f, {sin,cos}, a)])."""
    out = []
    for slot in range(ke):
        if slot < 6 * n_freq:
            pair, t = slot >> 1, slot & 1
            f, a = pair // 3, pair % 3
            out.append(3 + f * 6 + t * 3 + a)
        elif slot < 6 * n_freq + 3:
            out.append(slot - 6 * n_freq)
        else:
            out.append(-1)
    return out


class LayerBinding:
    """Which columns of the reference weight feed which kernel input block."""

    def __init__(self, wname, bname, emb0=None, cond=None, prev0=None, aux0=None):
        self.wname, self.bname = wname, bname
        self.aux0 = aux0    # first reference column of the 3 aux (view direction) channels riding in the embedding block (or None)
        self.emb0 = emb0    # first reference column of the embedding block (or None)
        self.cond = cond    # (first column, count) of the per-frame conditioning block (or None)
        self.prev0 = prev0  # first reference column of the previous-activation block (or None)


def bindings(net, prefix=""):
    p = prefix
    if net == NET_FG_BASE:  # nerf.py:99-109,134
        b = [LayerBinding(p + "basefield.linear_1.0.weight", p + "basefield.linear_1.0.bias", emb0=0, cond=(63, 32))]
        for i in (2, 3, 4):
            b.append(LayerBinding(p + f"basefield.linear_{i}.0.weight", p + f"basefield.linear_{i}.0.bias", prev0=0))
        b.append(LayerBinding(p + "basefield.linear_5.0.weight", p + "basefield.linear_5.0.bias", emb0=0, cond=(63, 32), prev0=95))
        for i in (6, 7, 8):
            b.append(LayerBinding(p + f"basefield.linear_{i}.0.weight", p + f"basefield.linear_{i}.0.bias", prev0=0))
        b.append(LayerBinding(p + "basefield.linear_final.0.weight", p + "basefield.linear_final.0.bias", prev0=0))
        b.append(LayerBinding(p + "sdf.weight", p + "sdf.bias", prev0=0))
        return b
    if net == NET_FG_COLOR:  # nerf.py:112-123,135-139
        return [LayerBinding(p + "colorfield.linear_1.0.weight", p + "colorfield.linear_1.0.bias", emb0=0, cond=(75, 32)),
                LayerBinding(p + "colorfield.linear_2.0.weight", p + "colorfield.linear_2.0.bias", prev0=0),
                LayerBinding(p + "colorfield.linear_final.0.weight", p + "colorfield.linear_final.0.bias", prev0=0),
                LayerBinding(p + "rgb.0.weight", p + "rgb.0.bias", prev0=0, cond=(256, 32)),
                LayerBinding(p + "rgb.2.weight", p + "rgb.2.bias", prev0=0)]
    if net == NET_VIS:  # visibility.py:39-51
        q = p + "vis_mlp.basefield."
        return [LayerBinding(q + "linear_1.0.weight", q + "linear_1.0.bias", emb0=0, cond=(63, 32)),
                LayerBinding(q + "linear_2.0.weight", q + "linear_2.0.bias", prev0=0),
                LayerBinding(q + "linear_final.weight", q + "linear_final.bias", prev0=0)]
    if net == NET_FEAT:  # feature.py:77-84
        q = p + "feature_field."
        b = [LayerBinding(q + "linear_1.0.weight", q + "linear_1.0.bias", emb0=0)]
        for i in (2, 3, 4):
            b.append(LayerBinding(q + f"linear_{i}.0.weight", q + f"linear_{i}.0.bias", prev0=0))
        b.append(LayerBinding(q + "linear_5.0.weight", q + "linear_5.0.bias", emb0=0, prev0=39))
        b.append(LayerBinding(q + "linear_final.weight", q + "linear_final.bias", prev0=0))
        return b
    if net in (NET_SKIN, NET_SKIN18):  # skinning.py:70-86: [3B bone coords | 128 time embedding | 32 instance code], B = 25 / 18
        q = p + "warp.skinning_model.delta_field."
        return [LayerBinding(q + "linear_1.0.weight", q + "linear_1.0.bias", emb0=0, cond=(75 if net == NET_SKIN else 54, 160)),
                LayerBinding(q + "linear_2.0.weight", q + "linear_2.0.bias", prev0=0),
                LayerBinding(q + "linear_final.weight", q + "linear_final.bias", prev0=0)]
    if net in (NET_SKIN_A, NET_SKIN18_A):  # the same module with linear_1 folded into the per-frame table (warping.skin_affine_table): layers = linear_2, linear_final
        q = p + "warp.skinning_model.delta_field."
        return [LayerBinding(q + "linear_2.0.weight", q + "linear_2.0.bias", emb0=0),
                LayerBinding(q + "linear_final.weight", q + "linear_final.bias", prev0=0)]
    if net == NET_DENSE:  # warping.py:123-141: [39 posenc | 128 time embedding | 32 instance code]; prefix selects the map,
        q = p  # "warp.post_warp.forward_map." / "warp.post_warp.backward_map." (the CondMLP itself, base.py:80-121)
        return [LayerBinding(q + "linear_1.0.weight", q + "linear_1.0.bias", emb0=0, cond=(39, 160)),
                LayerBinding(q + "linear_2.0.weight", q + "linear_2.0.bias", prev0=0),
                LayerBinding(q + "linear_final.weight", q + "linear_final.bias", prev0=0)]
    if net == NET_DENSE6:  # fg_motion "dense" (warping.py:94-141, class defaults D=6, skips=[4]): skip layer = [39 posenc | 128 time | 32 code | 256 previous]
        q = p
        b = [LayerBinding(q + "linear_1.0.weight", q + "linear_1.0.bias", emb0=0, cond=(39, 160))]
        for i in (2, 3, 4):
            b.append(LayerBinding(q + f"linear_{i}.0.weight", q + f"linear_{i}.0.bias", prev0=0))
        b.append(LayerBinding(q + "linear_5.0.weight", q + "linear_5.0.bias", emb0=0, cond=(39, 160), prev0=199))
        b.append(LayerBinding(q + "linear_6.0.weight", q + "linear_6.0.bias", prev0=0))
        b.append(LayerBinding(q + "linear_final.weight", q + "linear_final.bias", prev0=0))
        return b
    if net == NET_BG_BASE:  # multifields.py:86-93, nerf.py:95-109: [39 posenc | 32 instance code], D=5, skip at 4
        b = [LayerBinding(p + "basefield.linear_1.0.weight", p + "basefield.linear_1.0.bias", emb0=0, cond=(39, 32))]
        for i in (2, 3, 4):
            b.append(LayerBinding(p + f"basefield.linear_{i}.0.weight", p + f"basefield.linear_{i}.0.bias", prev0=0))
        b.append(LayerBinding(p + "basefield.linear_5.0.weight", p + "basefield.linear_5.0.bias", emb0=0, cond=(39, 32), prev0=71))
        b.append(LayerBinding(p + "basefield.linear_final.0.weight", p + "basefield.linear_final.0.bias", prev0=0))
        b.append(LayerBinding(p + "sdf.weight", p + "sdf.bias", prev0=0))
        return b
    if net == NET_BG_COLOR:  # nerf.py:112-139: [51 posenc | 32 code] ; rgb.0 input = [128 feature | 3 raw view direction]
        return [LayerBinding(p + "colorfield.linear_1.0.weight", p + "colorfield.linear_1.0.bias", emb0=0, cond=(51, 32)),
                LayerBinding(p + "colorfield.linear_2.0.weight", p + "colorfield.linear_2.0.bias", prev0=0),
                LayerBinding(p + "colorfield.linear_final.0.weight", p + "colorfield.linear_final.0.bias", prev0=0),
                LayerBinding(p + "rgb.0.weight", p + "rgb.0.bias", prev0=0, aux0=128),
                LayerBinding(p + "rgb.2.weight", p + "rgb.2.bias", prev0=0)]
    if net == NET_HASH_GEO:  # hashfield.py: raw 32 hash features -> 64 -> 16
        q = p + "hash.geo."
        return [LayerBinding(q + "0.weight", q + "0.bias", emb0=0), LayerBinding(q + "2.weight", q + "2.bias", prev0=0)]
    if net == NET_HASH_COLOR:  # hashfield.py: raw [16 geometry features | 3 view direction] -> 64 -> 64 -> 3
        q = p + "hash.color."
        return [LayerBinding(q + "0.weight", q + "0.bias", emb0=0), LayerBinding(q + "2.weight", q + "2.bias", prev0=0),
                LayerBinding(q + "4.weight", q + "4.bias", prev0=0)]
    raise ValueError(net)


def skin_net_for(n_bones, affine=False):
    """The delta-skin network instantiation of a skeleton: 25 bones (bob, skel-quad) or 18 (skel-human).  affine=True: the form with
    linear_1 folded into a per-frame (64 x 4) table of the point (NET_SKIN_A / NET_SKIN18_A)."""
    if n_bones == 25:
        return NET_SKIN_A if affine else NET_SKIN
    if n_bones == 18:
        return NET_SKIN18_A if affine else NET_SKIN18
    raise NotImplementedError("lab4d_amd: the delta-skin network is instantiated for 25 and 18 bones (got %d)" % n_bones)


_COLMAP = {}


def col_map(net, layer, device):
    """(ke+kin) int32: kernel input column -> column of the reference weight (or -1)."""
    key = (net, layer, str(device))
    if key not in _COLMAP:
        d = describe(net)
        L = d.layers[layer]
        bd = bindings(net)[layer]
        cm = []
        if L.ke:
            if d.emb_kind == 0:
                # posenc slots -> reference channels of this layer's embedding block (if it has one); slots 6L+3..6L+5 carry
                # the aux 3-vector of nets that have one (LAB4D_NET_BG_COLOR) and map to the layer's aux columns
                ref = posenc_slot_to_ref_channel(d.n_freq, L.ke)
                for slot in range(L.ke):
                    c = ref[slot]
                    a0 = 6 * d.n_freq + 3
                    if bd.emb0 is not None and c >= 0:
                        cm.append(bd.emb0 + c)
                    elif bd.aux0 is not None and a0 <= slot < a0 + 3:
                        cm.append(bd.aux0 + slot - a0)
                    else:
                        cm.append(-1)
            elif d.emb_kind == 2:  # slots = the hidden features of the folded first layer, in order
                cm += [bd.emb0 + c for c in range(L.ke)]
            else:
                cm += [(bd.emb0 + c if c < d.c_in else -1) for c in range(L.ke)]
        if L.kin:
            cm += [bd.prev0 + j for j in range(L.kin)]
        _COLMAP[key] = torch.tensor(cm, dtype=torch.int32, device=device)
    return _COLMAP[key]


# Fused gradient accumulation (the training loop's switch): when a weight / bias already HAS a .grad buffer (the optimizer's
# flat gradient views, lab4d_amd.optim.FlatAdamW), the weight-gradient kernels add into it directly, in reference layout
# (lab4d_mlp_wgrad_mapped), and autograd is handed None for that input -- no zero-fill, no column scatter, no AccumulateGrad add
# per layer and use.  Off (default): gradients are returned to autograd like any other Function.
FUSED_GRAD_ACCUM = False


FUSED_NARROW_BWD = os.environ.get("LAB4D_FUSED_NARROW", "1") != "0"  # 0: the stored-activation path for the narrow nets too (A/B measurements, parity tests of both)


def _grad_sink(p):
    g = p.grad if FUSED_GRAD_ACCUM else None
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape:
        return None
    return g


ALWAYS_PACK = False  # set while capturing a hipGraph: the pack kernels must be part of the graph (weights change between replays)
_COLIDX = {}


def col_index(net, layer, device):
    """(kernel columns, reference columns) LongTensors of the valid entries of col_map -- built once on the host so the
    scatter of a weight gradient back into reference layout needs no boolean indexing (no device->host sync)."""
    key = (net, layer, str(device))
    if key not in _COLIDX:
        cm = col_map(net, layer, device).cpu()
        k = torch.nonzero(cm >= 0).flatten()
        _COLIDX[key] = (k.to(device), cm[k].long().to(device))
    return _COLIDX[key]


_PACK_CACHE = {}  # id(weight tensor) -> (weakref to it, {(net, layer, prec, transposed): (version, packed)})


def packed_weights(net, layer, prec, W, transposed):
    """Pack one layer's weights for the chain kernels.  Cached per weight *object* and invalidated by its
    autograd version counter (an in-place optimizer step bumps it); never keyed on addresses, which the
    caching allocator recycles.

    INVARIANT for callers: a weight update must bump `W._version` -- every in-place torch op on the parameter and
    FlatAdamW.step (increment_version) do; writes through `p.data`, `.detach()` aliases or raw pointers do NOT, and must be
    followed by `clear_caches()` (or `repack_all(force=True)` under a captured graph)."""
    key = (net, layer, prec, transposed)
    ver = W._version
    ent = _PACK_CACHE.get(id(W))
    if ent is None or ent[0]() is not W:
        ent = (weakref.ref(W), {})
        _PACK_CACHE[id(W)] = ent
        weakref.finalize(W, _PACK_CACHE.pop, id(W), None)
    per = ent[1]
    hit = per.get(key)
    if hit is not None and hit[0] == ver and hit[1].device == W.device and not ALWAYS_PACK:
        return hit[1]
    out = _pack_into(net, layer, prec, W, transposed, hit[1] if (hit is not None and hit[1].device == W.device) else None)
    if not ALWAYS_PACK:
        per[key] = (ver, out)
    return out


def _pack_into(net, layer, prec, W, transposed, out=None):
    """Run the pack kernel; `out` (a previous packed buffer of the same layer) is overwritten in place so that its address
    -- which a captured hipGraph may hold -- stays valid across optimizer steps."""
    d = describe(net)
    L = d.layers[layer]
    Wc = W.detach().contiguous()
    _lib.require_device(Wc)
    if out is None or ALWAYS_PACK:
        out = torch.empty(L.mout_pad * (L.ke + L.kin), dtype=store_dtype(prec), device=W.device)
    cm = col_map(net, layer, W.device)
    _lib.check(_lib.lib().lab4d_mlp_pack(net, layer, prec, 1 if transposed else 0, _lib.ptr(Wc), Wc.shape[1], _lib.ptr(cm),
                                         _lib.ptr(out), _lib.stream()), "mlp_pack")
    return out


def repack_all(force=False):
    """Refresh, in place, every cached packed copy whose parameter changed since it was packed (call after an optimizer step when
    the kernels that consume the copies are replayed from a captured hipGraph and therefore never come through packed_weights).
    force=True repacks regardless of the version counters (after an update that did not bump them)."""
    n = 0
    for ref, per in list(_PACK_CACHE.values()):
        W = ref()
        if W is None:
            continue
        for key, (ver, buf) in list(per.items()):
            if force or ver != W._version:
                per[key] = (W._version, _pack_into(key[0], key[1], key[2], W, key[3], buf))
                n += 1
    return n


def clear_caches():
    _PACK_CACHE.clear()
    _COLMAP.clear()
    _COLIDX.clear()


def pf_bias_of(net, layer, W, cond):
    """Per-frame bias (M, mout) = cond (M, C) @ W[:, cond columns]^T  (CondMLP's appended code, base.py:139-146)."""
    c0, n = bindings(net)[layer].cond
    return cond @ W[:, c0:c0 + n].t()


def s_pad_of(S):
    """Samples padded to whole workgroups of the chain kernels: 4 waves x 64 samples (lab4d_mlp.h)."""
    return (S + 255) // 256 * 256


def buf_numel(F, S_pad):
    """Elements of a stored-activation buffer: (S_pad/64) blocks of F*64 (+128 skew) elements (mlp_kernels.hpp block_stride)."""
    return (S_pad // 64) * (F * 64 + 128)


def ld_of(S_pad, prec):
    """Row stride of the [feature][sample] buffers.  +4352 bytes per row (17 x 256 B): with a power-of-two
    stride every feature row of a tile lands in the same HBM channel (measured 1 TB/s instead of >4)."""
    return S_pad  # blocked [sample-block][feature][64] layout: no row-stride padding needed


class MlpChain(Function):
    """out (S, c_out) [, export] = net(x; weights), differentiable wrt x, ext, per-frame biases, weights."""

    @staticmethod
    def forward(ctx, net, prec, spf, x, ext, freq_w, export_layer, n_pf, x2, *rest, aff=None):
        global _TAP
        # aff (never passed through apply(); warping.SkinChain calls this body directly): the raw-input nets form their inputs in
        # the kernel from the (S,3) points x and the per-frame affine rows aff (M, c_in, 4) -- lab4d_mlp_fwd_args.aff
        d = describe(net)
        NL = d.n_layers
        pfs = list(rest[:n_pf])
        params = list(rest[n_pf:])
        assert len(params) == 2 * NL
        Ws, bs = params[0::2], params[1::2]
        x = x.contiguous()
        _lib.require_device(x)
        if x.dtype != torch.float32:
            raise RuntimeError("MlpChain: x must be fp32")
        S = x.shape[0]
        S_pad = s_pad_of(S)
        ld = ld_of(S_pad, prec)
        dev = x.device
        sdt = store_dtype(prec)
        need_grad = any(ctx.needs_input_grad)
        # narrow nets (<= 64 wide, bf16): the backward recomputes the forward and forms the weight gradients in registers (lab4d_mlp_backward_fused), so
        # the forward stores nothing -- it runs in inference mode also when gradients are wanted
        fused = bool(need_grad and FUSED_NARROW_BWD and export_layer < 0 and ext is None and x2 is None and _TAP is None
                     and _lib.lib().lab4d_mlp_fused_backward_supported(net, prec, int(spf)))
        store = need_grad and not fused
        # nothing but d/dx wanted (the eval path's normals, nerf.py:455-493): the sdf basefields then store their ReLU sign words and embedding only, and
        # their backward writes no dZ (38 GB per 8.4 M samples each that nobody would read)
        dx_only = bool(store and net in (NET_FG_BASE, NET_BG_BASE) and export_layer < 0 and ext is None and _TAP is None
                       and not any(ctx.needs_input_grad[i] for i in range(len(ctx.needs_input_grad)) if i != 3))
        a = FwdArgs()
        a.net, a.precision, a.S, a.S_pad, a.ld, a.spf = net, prec, S, S_pad, ld, int(spf)
        a.x = _lib.dp(x)
        if d.emb_kind == 2 and aff is None:
            raise RuntimeError("MlpChain: net %d needs the per-frame affine table aff" % net)
        if aff is not None:
            rows = d.ke if d.emb_kind == 2 else d.c_in
            if tuple(aff.shape[1:]) != (rows, 4) or x.shape[1] != 3 or aff.dtype != torch.float32 or not aff.is_contiguous():
                raise RuntimeError("MlpChain: aff must be a contiguous fp32 (M, %d, 4) table and x the (S,3) points" % rows)
            _lib.require_device(aff)
            a.aff = _lib.dp(aff)
        if x2 is not None:
            x2 = x2.contiguous().float()
            _lib.require_device(x2)
            a.x2 = _lib.dp(x2)
        if freq_w is not None:
            freq_w = freq_w.contiguous().float()
            a.freq_w = _lib.dp(freq_w)
        keep = [x2]
        acts = [None] * NL
        masks = [None] * NL
        pf_i = 0
        pf_used = [None] * NL
        pw_used, bias_used = [None] * NL, [None] * NL
        for l in range(NL):
            L = d.layers[l]
            pw = packed_weights(net, l, prec, Ws[l], False)
            a.W[l] = _lib.dp(pw)
            b = bs[l].detach().float()
            if b.numel() != L.mout_pad:
                b = torch.nn.functional.pad(b, (0, L.mout_pad - b.numel()))
            b = b.contiguous()
            a.bias[l] = _lib.dp(b)
            keep += [pw, b]
            pw_used[l], bias_used[l] = pw, b
            if L.pf_bias:
                pf = (pfs[pf_i].detach().float() + b[None]).contiguous()  # kernel contract: the per-frame table includes the bias
                pf_i += 1
                if pf.shape[1] != L.mout_pad:
                    raise RuntimeError("per-frame bias of layer %d must have %d columns" % (l, L.mout_pad))
                a.pf_bias[l] = _lib.dp(pf)
                pf_used[l] = pf
                keep.append(pf)
            if (store and not dx_only and l + 1 < NL) or l == export_layer:
                acts[l] = torch.empty(buf_numel(L.mout_pad, S_pad), dtype=sdt, device=dev)
                a.act[l] = _lib.dp(acts[l])
            if store and L.relu and l + 1 < NL:
                tile = 64 if prec == PREC_BF16 else 32
                masks[l] = torch.empty((S_pad // tile) * (L.mout_pad // 32) * 64, dtype=torch.int32, device=dev)
                a.mask[l] = _lib.dp(masks[l])
        emb = None
        if store:
            emb = torch.empty(buf_numel(d.ke, S_pad), dtype=sdt, device=dev)
            a.emb = _lib.dp(emb)
        if ext is not None:
            ext = ext.contiguous()
            if ext.dtype != sdt:
                raise RuntimeError("ext must be stored as %s" % sdt)
            a.ext = _lib.dp(ext)
        out = torch.empty(S, d.c_out, device=dev)
        a.out = _lib.dp(out)
        # algorithmic HBM bytes of this launch: every stored tensor written once, inputs read once
        nbytes = sum(t.numel() * t.element_size() for t in acts + masks + [emb, ext, out, x] if t is not None)
        with _lib.timed(chain_kernel_name("fwd", net, prec, dx_only) + ("" if store else " inference"), (2.0 * S * NET_MACS[net], float(nbytes))):
            _lib.check(_lib.lib().lab4d_mlp_forward(ctypes.byref(a), _lib.stream()), "mlp_forward")
        ctx.meta = (net, prec, int(spf), S, S_pad, ld, export_layer, n_pf, pf_used)
        ctx.acts, ctx.masks, ctx.emb, ctx.ext = acts, masks, emb, ext
        ctx.dx_only = dx_only
        # what the fused backward reads again: the points, the annealing window, the affine table, packed weights / padded biases as the forward took them
        ctx.fused = {"x": x, "freq_w": freq_w, "aff": aff, "W": pw_used, "bias": bias_used} if fused else None
        if _TAP is not None:  # run_chain(tap=...): the training-mode pass's ReLU sign words and stored embedding, for EikonalSdf (references, not copies)
            if need_grad:
                _TAP.update(net=net, prec=prec, S=S, S_pad=S_pad, masks=list(masks), emb=emb)
            _TAP = None
        ctx.params = params
        ctx.x_shape = x.shape if aff is None else (S, d.c_in)  # what d_x is the gradient of: the net's own inputs
        ctx.has_x2 = x2 is not None
        ctx.aff_in = (x, aff) if (d.emb_kind == 2 and need_grad) else None  # the backward chain takes the adjoint of the affine first layer itself
        if export_layer is not None and export_layer >= 0:
            # a separate tensor object over the same storage: returning ctx.acts[export_layer] itself would make the node own a
            # tensor whose grad_fn is the node -- a reference cycle that keeps every stored activation of the chunk alive until
            # the cyclic garbage collector runs (measured: +0.7 GiB per chunk at 128^2 x 32, 140 GiB per chunk at the bench size)
            return out, acts[export_layer].view(-1)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out, d_export=None):
        net, prec, spf, S, S_pad, ld, export_layer, n_pf, pf_used = ctx.meta
        d = describe(net)
        NL = d.n_layers
        params = ctx.params
        Ws, bs = params[0::2], params[1::2]
        dev = d_out.device
        sdt = store_dtype(prec)
        if getattr(ctx, "fused", None) is not None:
            return MlpChain._backward_fused(ctx, d_out, d, Ws, bs)
        a = BwdArgs()
        a.net, a.precision, a.S, a.S_pad, a.ld, a.spf = net, prec, S, S_pad, ld, spf
        keep = []
        dz = [None] * NL
        for l in range(NL):
            L = d.layers[l]
            pw = packed_weights(net, l, prec, Ws[l], True)
            a.WT[l] = _lib.dp(pw)
            keep.append(pw)
            if ctx.acts[l] is not None:
                a.act[l] = _lib.dp(ctx.acts[l])
            if ctx.masks[l] is not None:
                a.mask[l] = _lib.dp(ctx.masks[l])
            if not getattr(ctx, "dx_only", False):
                dz[l] = torch.empty(buf_numel(L.mout_pad, S_pad), dtype=sdt, device=dev)
                a.dz[l] = _lib.dp(dz[l])
            if L.ext_grad:
                if d_export is None:
                    d_export = torch.zeros(buf_numel(L.mout_pad, S_pad), dtype=sdt, device=dev)
                d_export = d_export.contiguous()
                a.ext_gin = _lib.dp(d_export)
        if ctx.emb is not None:
            a.emb = _lib.dp(ctx.emb)
        ext_g = None
        if ctx.ext is not None:
            a.ext = _lib.dp(ctx.ext)
            ext_g = torch.empty_like(ctx.ext)  # always written by the kernel (no stores in runtime branches)
            a.ext_gout = _lib.dp(ext_g)
        d_out = d_out.contiguous().float()
        a.d_out = _lib.dp(d_out)
        d_x = None
        d_x2 = None
        g_aff = None
        if ctx.aff_in is not None:
            xin, aff = ctx.aff_in
            a.x, a.aff = _lib.dp(xin), _lib.dp(aff)
            g_aff = torch.zeros_like(aff)
            a.g_aff = _lib.dp(g_aff)
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[8] or g_aff is not None:
            d_x = torch.empty(ctx.x_shape, device=dev)
            a.d_x = _lib.dp(d_x)
            if ctx.has_x2:  # written together with d_x by the kernel
                d_x2 = torch.empty(ctx.x_shape, device=dev)
                a.d_x2 = _lib.dp(d_x2)
        # algorithmic HBM bytes: every dZ written once; masks, head gradient, stored embedding / external tensors read once
        nbytes = sum(t.numel() * t.element_size() for t in list(dz) + list(ctx.masks) + [d_out, d_x, ext_g, ctx.emb if d_x is not None else None]
                     if t is not None)
        if d_export is not None:
            nbytes += d_export.numel() * d_export.element_size()
        with _lib.timed(chain_kernel_name("bwd", net, prec, getattr(ctx, "dx_only", False)), (2.0 * S * NET_MACS[net], float(nbytes))):
            _lib.check(_lib.lib().lab4d_mlp_backward(ctypes.byref(a), _lib.stream()), "mlp_backward")
        # weight / bias gradients
        M = (S + spf - 1) // spf
        grads_pf, grads_params = [], []
        # one zero-filled arena for every accumulated output of the wgrad launches (one fill instead of ~3 per layer); layers
        # whose weight has a gradient sink (FUSED_GRAD_ACCUM) need no scratch matrix at all
        sizes, sinks = [], []
        for l in range(NL):
            L = d.layers[l]
            need_w = ctx.needs_input_grad[9 + n_pf + 2 * l]
            need_b = ctx.needs_input_grad[9 + n_pf + 2 * l + 1]
            sw = _grad_sink(Ws[l]) if need_w else None
            sb = _grad_sink(bs[l]) if (need_b and sw is not None and not L.pf_bias) else None
            sinks.append((sw, sb))
            sizes.append((0 if sw is not None else L.mout_pad * (L.ke + L.kin), 0 if sb is not None else L.mout_pad,
                          M * L.mout_pad if L.pf_bias else 0))
        arena = torch.zeros(sum(sum(t) for t in sizes), device=dev)
        aoff = 0
        pf_seen = 0
        for l in range(NL):
            L = d.layers[l]
            K = L.ke + L.kin
            need_w = ctx.needs_input_grad[9 + n_pf + 2 * l]
            need_b = ctx.needs_input_grad[9 + n_pf + 2 * l + 1]
            # the per-frame bias table is an input of its own (position 9 + its rank among the pf layers): when nothing upstream of it wants a gradient
            # (the eval path's normals differentiate wrt the points only) its layer needs no weight-gradient launch at all
            need_pf = bool(L.pf_bias) and bool(ctx.needs_input_grad[9 + pf_seen])
            pf_seen += 1 if L.pf_bias else 0
            gW = gb = None
            if need_w or need_b or need_pf:
                n0, n1, n2 = sizes[l]
                sw, sb = sinks[l]
                dWk = arena[aoff:aoff + n0].view(L.mout_pad, K) if sw is None else None
                dbk = arena[aoff + n0:aoff + n0 + n1] if sb is None else None
                pfd = arena[aoff + n0 + n1:aoff + n0 + n1 + n2].view(M, L.mout_pad) if need_pf else None
                prev = ctx.acts[l - 1] if L.kin else None
                with _lib.timed(wgrad_kernel_name(L, prec), wgrad_work(L, S_pad, prec)):
                    if sw is not None:
                        _lib.check(_lib.lib().lab4d_mlp_wgrad_mapped(net, l, prec, S, S_pad, ld, spf, _lib.ptr(dz[l]), _lib.ptr(ctx.emb), _lib.ptr(prev),
                                                                     _lib.ptr(sw), sw.shape[1], _lib.ptr(col_map(net, l, dev)),
                                                                     _lib.ptr(sb if sb is not None else dbk), _lib.ptr(pfd), M, _lib.stream()),
                                   "mlp_wgrad_mapped")
                    else:
                        _lib.check(_lib.lib().lab4d_mlp_wgrad(net, l, prec, S, S_pad, ld, spf, _lib.ptr(dz[l]), _lib.ptr(ctx.emb), _lib.ptr(prev),
                                                              _lib.ptr(dWk), _lib.ptr(dbk), _lib.ptr(pfd), M, _lib.stream()), "mlp_wgrad")
                if need_w and sw is None:
                    kcols, rcols = col_index(net, l, dev)
                    gW = torch.zeros_like(Ws[l], dtype=torch.float32)
                    gW[:, rcols] = dWk[:L.mout][:, kcols]
                if need_b and sb is None:
                    gb = (pfd.sum(0) if need_pf else dbk)[:L.mout].reshape(bs[l].shape)
                if need_pf:
                    grads_pf.append(pfd)
            if L.pf_bias and not need_pf:
                grads_pf.append(None)
            aoff += sum(sizes[l])
            grads_params += [gW, gb]
        # Release the stored activations / masks / embedding NOW.  They are plain attributes of ctx (not save_for_backward tensors),
        # so autograd would keep them until the whole graph dies at the end of backward(): every net's activations stayed alive
        # through every other net's backward (measured: backward peak = everything the forward saved + the largest dZ set, 15.7 KB
        # per sample).  Like freed saved tensors, this makes a second backward through the node an error.
        ctx.acts = ctx.masks = ctx.emb = ctx.ext = ctx.params = ctx.aff_in = None
        ctx.g_aff = g_aff  # read by the caller that passed aff (warping.SkinChainA)
        return (None, None, None, d_x, ext_g, None, None, None, d_x2, *grads_pf, *grads_params)

    @staticmethod
    def _backward_fused(ctx, d_out, d, Ws, bs):
        """The narrow nets' backward in ONE launch (lab4d_mlp_backward_fused, csrc/mlp_fused_bwd.hpp): recompute + dgrad chain + every layer's weight /
        bias gradient in registers.  Same return tuple as backward()."""
        net, prec, spf, S, S_pad, ld, export_layer, n_pf, pf_used = ctx.meta
        NL = d.n_layers
        f = ctx.fused
        dev = d_out.device
        M = (S + spf - 1) // spf
        a = BwdFusedArgs()
        a.net, a.precision, a.S, a.spf = net, prec, S, spf
        a.x = _lib.dp(f["x"])
        if f["freq_w"] is not None:
            a.freq_w = _lib.dp(f["freq_w"])
        g_aff = None
        if f["aff"] is not None:
            a.aff = _lib.dp(f["aff"])
            g_aff = torch.zeros_like(f["aff"])
            a.g_aff = _lib.dp(g_aff)
        d_out = d_out.contiguous().float()
        a.d_out = _lib.dp(d_out)
        d_x = None
        if ctx.needs_input_grad[3] or g_aff is not None:
            d_x = torch.empty(ctx.x_shape, device=dev)
            a.d_x = _lib.dp(d_x)
        # one zero-filled arena for every accumulated output: dW (mout_pad, 64) in kernel column order, db (mout_pad), pf_db (M, mout_pad)
        sizes = [(L.mout_pad * (L.ke + L.kin), L.mout_pad, M * L.mout_pad if L.pf_bias else 0) for L in (d.layers[l] for l in range(NL))]
        arena = torch.zeros(sum(sum(t) for t in sizes), device=dev)
        keep, views, aoff = [], [], 0
        for l in range(NL):
            L = d.layers[l]
            n0, n1, n2 = sizes[l]
            dWk, dbk = arena[aoff:aoff + n0].view(L.mout_pad, L.ke + L.kin), arena[aoff + n0:aoff + n0 + n1]
            pfd = arena[aoff + n0 + n1:aoff + n0 + n1 + n2].view(M, L.mout_pad) if L.pf_bias else None
            aoff += n0 + n1 + n2
            views.append((dWk, dbk, pfd))
            pwt = packed_weights(net, l, prec, Ws[l], True)
            keep.append(pwt)
            a.WT[l], a.dW[l] = _lib.dp(pwt), _lib.dp(dWk)
            if l + 1 < NL:
                a.W[l] = _lib.dp(f["W"][l])
            if L.pf_bias:
                a.pf_bias[l], a.pf_db[l] = _lib.dp(pf_used[l]), _lib.dp(pfd)
            else:
                a.bias[l], a.db[l] = _lib.dp(f["bias"][l]), _lib.dp(dbk)
        nbytes = 4.0 * S * (3 + d.c_out + (3 if d_x is not None else 0))
        with _lib.timed("k_mlp_bwd_fused<%s>" % KERNEL_NET[net], (2.0 * S * NET_MACS[net] * 3, nbytes)):  # recompute + dgrad + wgrad
            _lib.check(_lib.lib().lab4d_mlp_backward_fused(ctypes.byref(a), _lib.stream()), "mlp_backward_fused")
        grads_pf, grads_params = [], []
        pf_seen = 0
        for l in range(NL):
            L = d.layers[l]
            dWk, dbk, pfd = views[l]
            need_w = ctx.needs_input_grad[9 + n_pf + 2 * l]
            need_b = ctx.needs_input_grad[9 + n_pf + 2 * l + 1]
            gW = gb = None
            if need_w:
                kcols, rcols = col_index(net, l, dev)
                sw = _grad_sink(Ws[l])
                if sw is not None:  # FUSED_GRAD_ACCUM: straight into weight.grad (a view of the optimizer's flat buffer)
                    sw.index_add_(1, rcols, dWk[:L.mout].index_select(1, kcols))
                else:
                    gW = torch.zeros_like(Ws[l], dtype=torch.float32)
                    gW[:, rcols] = dWk[:L.mout][:, kcols]
            if need_b:
                gvec = (pfd.sum(0) if L.pf_bias else dbk)[:L.mout].reshape(bs[l].shape)
                sb = _grad_sink(bs[l]) if need_w and _grad_sink(Ws[l]) is not None else None
                if sb is not None:
                    sb.add_(gvec)
                else:
                    gb = gvec
            if L.pf_bias:
                grads_pf.append(pfd if ctx.needs_input_grad[9 + pf_seen] else None)
                pf_seen += 1
            grads_params += [gW, gb]
        ctx.fused = ctx.params = ctx.aff_in = None
        ctx.g_aff = g_aff
        return (None, None, None, d_x if ctx.needs_input_grad[3] or g_aff is not None else None, None, None, None, None, None, *grads_pf, *grads_params)



_TAP = None


def run_chain(net, prec, P, x, spf, conds=None, ext=None, freq_w=None, export_layer=None, prefix="", x2=None, pfs_pre=None, tap=None):
    """Convenience wrapper: P maps reference state_dict names -> device tensors; conds maps layer index ->
    (M, C) per-frame conditioning input of that layer.  pfs_pre maps layer index -> an already evaluated per-frame bias
    pf_bias_of(net, l, W, conds[l]) (the per-frame prologue of a training step, deformable.frame_terms: the table does not
    depend on the rays, so it is formed once per step, not once per chunk).  Returns out or (out, exported activation).
    tap: a dict that receives this (training-mode) pass's stored ReLU sign words and embedding -- eikonal_sdf(tap=...) then takes its primal
    pattern from them instead of running the primal forward again on its subset of the same samples."""
    global _TAP
    _TAP = tap
    d = describe(net)
    bd = bindings(net, prefix)
    pfs = []
    for l in range(d.n_layers):
        if d.layers[l].pf_bias:
            pfs.append(pfs_pre[l] if pfs_pre is not None and l in pfs_pre else pf_bias_of(net, l, P[bd[l].wname], conds[l]))
    params = []
    for l in range(d.n_layers):
        params += [P[bd[l].wname], P[bd[l].bname]]
    return MlpChain.apply(net, prec, spf, x, ext, freq_w, -1 if export_layer is None else export_layer, len(pfs), x2, *pfs, *params)



@torch.no_grad()
def run_chain_compacted(net, prec, P, x, frame_idx, count, conds=None, ext=None, freq_w=None, export_layer=None, prefix=""):
    """Inference-mode chain on a stream-compacted sample list (NeRF.query_nerf, nerf.py:782-808): x (S,3) holds the valid samples
    in its first *count rows (count: int32 device scalar -- it never visits the host), frame_idx (S) int32 names the frame each
    of them belongs to (compacted samples are not frame-contiguous, so `s // spf` no longer works; the per-frame bias tables keep
    their M rows instead of being expanded per sample as nerf.py:795-798 does).  Tiles beyond the count are skipped by the kernel;
    rows >= *count of the result are not written.  Returns out or (out, exported activation)."""
    d = describe(net)
    bd = bindings(net, prefix)
    x = x.contiguous()
    _lib.require_device(x, frame_idx, count)
    if frame_idx.dtype != torch.int32 or count.dtype != torch.int32:
        raise RuntimeError("run_chain_compacted: frame_idx and count must be int32")
    S = x.shape[0]
    S_pad = s_pad_of(S)
    dev, sdt = x.device, store_dtype(prec)
    a = FwdArgs()
    a.net, a.precision, a.S, a.S_pad, a.ld, a.spf = net, prec, S, S_pad, S_pad, 1
    a.x, a.S_dev, a.frame_idx = _lib.dp(x), _lib.dp(count), _lib.dp(frame_idx)
    keep = []
    if freq_w is not None:
        freq_w = freq_w.contiguous().float()
        a.freq_w = _lib.dp(freq_w)
    exported = None
    for l in range(d.n_layers):
        L = d.layers[l]
        W, b = P[bd[l].wname], P[bd[l].bname].detach().float()
        pw = packed_weights(net, l, prec, W, False)
        a.W[l] = _lib.dp(pw)
        if b.numel() != L.mout_pad:
            b = torch.nn.functional.pad(b, (0, L.mout_pad - b.numel()))
        b = b.contiguous()
        a.bias[l] = _lib.dp(b)
        keep += [pw, b]
        if L.pf_bias:
            pf = (pf_bias_of(net, l, W, conds[l]).float() + b[None]).contiguous()
            a.pf_bias[l] = _lib.dp(pf)
            keep.append(pf)
        if l == export_layer:
            exported = torch.empty(buf_numel(L.mout_pad, S_pad), dtype=sdt, device=dev)
            a.act[l] = _lib.dp(exported)
    if ext is not None:
        if ext.dtype != sdt:
            raise RuntimeError("ext must be stored as %s" % sdt)
        a.ext = _lib.dp(ext)
    out = torch.empty(S, d.c_out, device=dev)
    a.out = _lib.dp(out)
    with _lib.timed(chain_kernel_name("fwd", net, prec) + " inference", (2.0 * S * NET_MACS[net], 0.0)):
        _lib.check(_lib.lib().lab4d_mlp_forward(ctypes.byref(a), _lib.stream()), "mlp_forward(compacted)")
    return (out, exported) if exported is not None else out


class EikonalSdf(Function):
    """e[s] = (|d sdf / d x_s| - 1)^2 for detached points x (S,3): NeRF.compute_eikonal / torch_utils.compute_gradient
    (nerf.py:416-453, torch_utils.py:4-27) without second-order autograd.  Forward = primal chain + dgrad chain with
    d_out = 1 (gives g = d sdf/dx and the backward signals dz_l).  Backward = tangent-mode forward of u = J_e(x) dL/dg
    through the same ReLU pattern, then the ordinary wgrad kernel on (dz_l, tangent activations): see
    lab4d_mlp_forward_tangent in include/lab4d_mlp.h for the derivation."""

    @staticmethod
    def forward(ctx, net, prec, spf, x, freq_w, pf0, pf4, *params):
        global _EIK_TAP
        tap, _EIK_TAP = _EIK_TAP, None
        if net not in (NET_FG_BASE, NET_BG_BASE):
            raise RuntimeError("EikonalSdf: the eikonal term exists for the basefield / sdf networks only (net %d)" % net)
        d = describe(net)
        NL = d.n_layers
        Ws, bs = params[0::2], params[1::2]
        x = x.detach().contiguous()
        _lib.require_device(x)
        S = x.shape[0]
        S_pad = s_pad_of(S)
        dev, sdt = x.device, store_dtype(prec)
        tile = 64 if prec == PREC_BF16 else 32
        a = FwdArgs()
        a.net, a.precision, a.S, a.S_pad, a.ld, a.spf = net, prec, S, S_pad, S_pad, int(spf)
        a.x = _lib.dp(x)
        fw = None
        if freq_w is not None:
            fw = freq_w.detach().contiguous().float()
            a.freq_w = _lib.dp(fw)
        keep, masks, packed = [], [None] * NL, []
        pfs = {0: pf0.detach().contiguous().float(), 4: pf4.detach().contiguous().float()}
        for l in range(NL):
            L = d.layers[l]
            pw = packed_weights(net, l, prec, Ws[l], False)
            packed.append(pw)
            a.W[l] = _lib.dp(pw)
            b = bs[l].detach().float()
            if b.numel() != L.mout_pad:
                b = torch.nn.functional.pad(b, (0, L.mout_pad - b.numel()))
            b = b.contiguous()
            keep.append(b)
            a.bias[l] = _lib.dp(b)
            if L.pf_bias:
                pfb = (pfs[l].detach().float() + b[None]).contiguous()
                keep.append(pfb)
                a.pf_bias[l] = _lib.dp(pfb)
            if L.relu and l + 1 < NL:
                masks[l] = torch.empty((S_pad // tile) * (L.mout_pad // 32) * 64, dtype=torch.int32, device=dev)
                a.mask[l] = _lib.dp(masks[l])
        emb = torch.empty(buf_numel(d.ke, S_pad), dtype=sdt, device=dev)
        a.emb = _lib.dp(emb)
        # training-mode forward stores every hidden activation; the primal ones are not needed here, so the buffers are
        # the ones the tangent pass of backward() overwrites with the tangent activations
        tact = [None] * NL
        for l in range(NL - 1):
            tact[l] = torch.empty(buf_numel(d.layers[l].mout_pad, S_pad), dtype=sdt, device=dev)
            a.act[l] = _lib.dp(tact[l])
        sdf = torch.empty(S, 1, device=dev)
        a.out = _lib.dp(sdf)
        reused = False
        if tap is not None:
            # The points are whole 64-sample blocks of a training-mode pass of this very network that has just run (same weights, per-frame
            # biases and annealing window: deformable.query_field_train evaluates the field on ALL samples and the eikonal term on a drawn 1/16
            # of the rays): the primal pass's ReLU sign words and stored embedding are GATHERED from that pass's buffers (rows = tiles / blocks)
            # instead of being recomputed by a forward launch of their own.
            src, blk_map = tap  # blk_map: (S / 64) int64, 64-sample block of the tapped pass that block i of x is
            ok = (src.get("net") == net and src.get("prec") == prec and S % 64 == 0 and S_pad == S and blk_map.numel() == S // 64
                  and src.get("emb") is not None and all((masks[l] is None) == (src["masks"][l] is None) for l in range(NL)))
            if ok:
                nb_src = src["S_pad"] // 64
                tpb = 64 // tile  # tiles per block: 1 (bf16) / 2 (fp32)
                tile_map = blk_map if tpb == 1 else (blk_map[:, None] * tpb + torch.arange(tpb, device=dev)).reshape(-1)
                for l in range(NL):
                    if masks[l] is not None:
                        masks[l] = src["masks"][l].view(nb_src * tpb, -1).index_select(0, tile_map).reshape(-1)
                emb = src["emb"].view(nb_src, -1).index_select(0, blk_map).reshape(-1)
                reused = True
        if not reused:
            with _lib.timed(chain_kernel_name("fwd", net, prec) + "@eik"):
                _lib.check(_lib.lib().lab4d_mlp_forward(ctypes.byref(a), _lib.stream()), "mlp_forward(eikonal primal)")
        bk = BwdArgs()
        bk.net, bk.precision, bk.S, bk.S_pad, bk.ld, bk.spf = net, prec, S, S_pad, S_pad, int(spf)
        dz = [None] * NL
        for l in range(NL):
            L = d.layers[l]
            pt = packed_weights(net, l, prec, Ws[l], True)
            keep.append(pt)
            bk.WT[l] = _lib.dp(pt)
            if masks[l] is not None:
                bk.mask[l] = _lib.dp(masks[l])
            dz[l] = torch.empty(buf_numel(L.mout_pad, S_pad), dtype=sdt, device=dev)
            bk.dz[l] = _lib.dp(dz[l])
            if L.ext_grad:
                zg = torch.zeros(buf_numel(L.mout_pad, S_pad), dtype=sdt, device=dev)
                keep.append(zg)
                bk.ext_gin = _lib.dp(zg)
        bk.emb = _lib.dp(emb)
        ones = torch.ones(S, 1, device=dev)
        bk.d_out = _lib.dp(ones)
        g = torch.empty(S, 3, device=dev)
        bk.d_x = _lib.dp(g)
        with _lib.timed(chain_kernel_name("bwd", net, prec) + "@eik"):
            _lib.check(_lib.lib().lab4d_mlp_backward(ctypes.byref(bk), _lib.stream()), "mlp_backward(eikonal primal)")
        gn = g.norm(2, dim=-1, keepdim=True)
        ctx.meta = (net, prec, int(spf), S, S_pad)
        ctx.saved = (x, fw, g, gn, dz, masks, packed, tact)
        ctx.params = params
        return (gn - 1) ** 2

    @staticmethod
    @once_differentiable
    def backward(ctx, ge):
        net, prec, spf, S, S_pad = ctx.meta
        x, fw, g, gn, dz, masks, packed, tact = ctx.saved
        d = describe(net)
        NL, L0 = d.n_layers, d.n_freq
        Ws = ctx.params[0::2]
        dev, sdt = x.device, store_dtype(prec)
        # dL/dg, then u = J_e(x) dL/dg in embedding-slot order [ (f, a, {sin,cos}) pairs | x | pad ]
        # zero sdf gradient (every unit of a layer dead): torch's norm backward takes the zero subgradient there, not 0/0
        u = torch.empty(S, d.ke, device=dev)
        _lib.check(_lib.lib().lab4d_eikonal_tangent_input(_lib.ptr(x), _lib.ptr(g), _lib.ptr(ge.contiguous().float()), _lib.ptr(fw), S, L0, d.ke, _lib.ptr(u),
                                                          _lib.stream()), "eikonal_tangent_input")
        a = FwdArgs()
        a.net, a.precision, a.S, a.S_pad, a.ld, a.spf = net, prec, S, S_pad, S_pad, spf
        a.x = _lib.dp(u)
        for l in range(NL):
            L = d.layers[l]
            a.W[l] = _lib.dp(packed[l])
            if masks[l] is not None:
                a.mask[l] = _lib.dp(masks[l])
            if l + 1 < NL:
                a.act[l] = _lib.dp(tact[l])
        temb = torch.empty(buf_numel(d.ke, S_pad), dtype=sdt, device=dev)
        a.emb = _lib.dp(temb)
        with _lib.timed(("k_mlp_fwd_ws_tangent<%s>@eik" if (ws_active(net, prec) and net == NET_FG_BASE) else "k_mlp_fwd_tangent<%s>@eik") % KERNEL_NET[net]):
            _lib.check(_lib.lib().lab4d_mlp_forward_tangent(ctypes.byref(a), _lib.stream()), "mlp_forward_tangent")
        sinks = [(_grad_sink(Ws[l]) if ctx.needs_input_grad[7 + 2 * l] else None) for l in range(NL)]
        sizes = [0 if sinks[l] is not None else d.layers[l].mout_pad * (d.layers[l].ke + d.layers[l].kin) for l in range(NL)]
        arena = torch.zeros(sum(sizes), device=dev)
        off = 0
        grads = []
        for l in range(NL):
            L = d.layers[l]
            gW = None
            if ctx.needs_input_grad[7 + 2 * l]:
                prev = tact[l - 1] if L.kin else None
                with _lib.timed(wgrad_kernel_name(L, prec) + "@eik", wgrad_work(L, S_pad, prec)):
                    if sinks[l] is not None:
                        _lib.check(_lib.lib().lab4d_mlp_wgrad_mapped(net, l, prec, S, S_pad, S_pad, spf, _lib.ptr(dz[l]), _lib.ptr(temb), _lib.ptr(prev),
                                                                     _lib.ptr(sinks[l]), sinks[l].shape[1], _lib.ptr(col_map(net, l, dev)), None, None, 0,
                                                                     _lib.stream()), "mlp_wgrad_mapped(eikonal)")
                    else:
                        dWk = arena[off:off + sizes[l]].view(L.mout_pad, L.ke + L.kin)
                        _lib.check(_lib.lib().lab4d_mlp_wgrad(net, l, prec, S, S_pad, S_pad, spf, _lib.ptr(dz[l]), _lib.ptr(temb), _lib.ptr(prev),
                                                              _lib.ptr(dWk), None, None, 0, _lib.stream()), "mlp_wgrad(eikonal)")
                if sinks[l] is None:
                    kcols, rcols = col_index(net, l, dev)
                    gW = torch.zeros_like(Ws[l], dtype=torch.float32)
                    gW[:, rcols] = dWk[:L.mout][:, kcols]
            off += sizes[l]
            grads += [gW, None]
        ctx.saved = None  # release the tangent pass's stored tensors now (see MlpChain.backward)
        return (None, None, None, None, None, None, None, *grads)


_EIK_TAP = None


def eikonal_sdf(P, x, ray_code, spf, prec, freq_w=None, prefix="", net=NET_FG_BASE, pf_rows=None, tap=None):
    """(|d sdf/dx| - 1)^2 at detached points x (S,3); ray_code (S/spf, 32) = instance code of the ray each group of `spf`
    consecutive samples belongs to.  net = NET_FG_BASE or NET_BG_BASE (both condition layers 0 and 4 on the code).
    pf_rows = (pf0, pf4) already evaluated per ray (rows of the per-frame tables, deformable.frame_terms) replaces ray_code.
    tap = (dict filled by run_chain(tap=...) of the SAME network on a superset of these samples, (S/64) int64 block map): the primal pass's
    ReLU pattern and embedding are taken from that pass."""
    bd = bindings(net, prefix)
    if pf_rows is not None:
        pf0, pf4 = pf_rows
    else:
        pf0 = pf_bias_of(net, 0, P[bd[0].wname], ray_code)
        pf4 = pf_bias_of(net, 4, P[bd[4].wname], ray_code)
    params = []
    for l in range(describe(net).n_layers):
        params += [P[bd[l].wname], P[bd[l].bname]]
    global _EIK_TAP
    _EIK_TAP = tap  # (tapped pass, block map): see EikonalSdf.forward
    return EikonalSdf.apply(net, prec, spf, x, freq_w, pf0, pf4, *params)
