"""Host-side mirror of lab4d/utils/render_utils.py -- same function names and signatures
(sample_cam_rays, render_pixel, compute_weights, integrate, sample_pdf), executed by the
gfx950 kernels of liblab4d_hip.so (csrc/raymarch.hip, csrc/composite.hip)."""
import ctypes

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib

KEY_SKIP = ["density", "vis", "flow", "eikonal", "xy_reproj", "xyz_reproj", "gauss_density"]  # render_utils.py:138-146
KEY_FREEZE = ["cyc_dist", "xyz_cam", "skin_entropy"]  # render_utils.py:147
KEY_MEAN = ["eikonal", "delta_skin"]  # render_utils.py:75-80


def _f32(*ts):
    _lib.require_device(*ts)
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError("this op is fp32 only (the reference path is fp32, SURVEY F6); got %s" % t.dtype)


class _RaySamples(Function):
    """sample_cam_rays (render_utils.py:8-56, perturb=False) fused with NeRF.cam_to_field
    (nerf.py:821-844).  Differentiable wrt Kinv and the cam->field rigid transform."""

    @staticmethod
    def forward(ctx, hxy, Kinv, near_far, depth_in, cq, ct, D):
        hxy, Kinv = hxy.contiguous(), Kinv.contiguous()
        near_far = near_far.contiguous() if near_far is not None else None
        depth_in = depth_in.contiguous() if depth_in is not None else None
        cq = cq.contiguous() if cq is not None else None
        ct = ct.contiguous() if ct is not None else None
        _f32(hxy, Kinv, near_far, depth_in, cq, ct)
        M, N = hxy.shape[:2]
        if depth_in is not None:
            D = depth_in.shape[2]
        dev = hxy.device
        xyz_cam = torch.empty(M, N, D, 3, device=dev)
        dir_cam = torch.empty(M, N, D, 3, device=dev)
        deltas = torch.empty(M, N, D, 1, device=dev)
        depth = torch.empty(M, N, D, 1, device=dev)
        xyz_f = torch.empty(M, N, D, 3, device=dev) if cq is not None else None
        dir_f = torch.empty(M, N, D, 3, device=dev) if cq is not None else None
        _lib.check(_lib.lib().lab4d_ray_samples_forward(
            _lib.ptr(hxy), _lib.ptr(Kinv), _lib.ptr(near_far), _lib.ptr(depth_in), _lib.ptr(cq), _lib.ptr(ct), M, N, D,
            _lib.ptr(xyz_cam), _lib.ptr(dir_cam), _lib.ptr(deltas), _lib.ptr(depth), _lib.ptr(xyz_f), _lib.ptr(dir_f),
            _lib.stream()), "ray_samples_forward")
        ctx.save_for_backward(hxy, Kinv, near_far, depth_in, cq, ct)
        ctx.dims = (M, N, D)
        ctx.mark_non_differentiable(depth)
        if cq is None:
            return xyz_cam, dir_cam, deltas, depth
        return xyz_cam, dir_cam, deltas, depth, xyz_f, dir_f

    @staticmethod
    @once_differentiable
    def backward(ctx, g_xyz, g_dir, g_deltas, g_depth, g_xyzf=None, g_dirf=None):
        hxy, Kinv, near_far, depth_in, cq, ct = ctx.saved_tensors
        M, N, D = ctx.dims
        dev = hxy.device

        def c(g):
            return g.contiguous() if g is not None else None

        g_xyz, g_dir, g_deltas, g_xyzf, g_dirf = c(g_xyz), c(g_dir), c(g_deltas), c(g_xyzf), c(g_dirf)
        gK = torch.zeros(M, 3, 3, device=dev)
        gq = torch.zeros(M, 4, device=dev) if cq is not None else None
        gt = torch.zeros(M, 3, device=dev) if cq is not None else None
        _lib.check(_lib.lib().lab4d_ray_samples_backward(
            _lib.ptr(hxy), _lib.ptr(Kinv), _lib.ptr(near_far), _lib.ptr(depth_in), _lib.ptr(cq), _lib.ptr(ct), M, N, D,
            _lib.ptr(g_xyz), _lib.ptr(g_dir), _lib.ptr(g_deltas), _lib.ptr(g_xyzf), _lib.ptr(g_dirf), _lib.ptr(gK),
            _lib.ptr(gq), _lib.ptr(gt), _lib.stream()), "ray_samples_backward")
        return None, gK, None, None, gq, gt, None


def sample_cam_rays(hxy, Kinv, near_far, n_depth=64, depth=None, perturb=False):
    """Same contract as render_utils.sample_cam_rays (render_utils.py:8-56).  perturb=True (stratified jitter,
    render_utils.py:36-42; no in-tree caller enables it): the jittered depths are formed from one torch.rand draw of
    the reference's shape on the rays' device -- (M,N,D,1) element-wise work -- and handed to the kernel through its
    `depth=` input, so rays, deltas and directions come from the same kernel either way."""
    if perturb:
        M, N = hxy.shape[:2]
        if depth is None:
            z = torch.linspace(0, 1, n_depth, device=hxy.device)[None]
            depth = (near_far[:, 0:1] * (1 - z) + near_far[:, 1:2] * z)[:, None, :, None].repeat(1, N, 1, 1)
        mid = 0.5 * (depth[:, :, :-1] + depth[:, :, 1:])
        upper = torch.cat([mid, depth[:, :, -1:]], -2)
        lower = torch.cat([depth[:, :, :1], mid], -2)
        depth = (lower + (upper - lower) * torch.rand(depth.shape, device=hxy.device)).contiguous()
    return _RaySamples.apply(hxy, Kinv, near_far, depth, None, None, n_depth)[:4]


def ray_samples(hxy, Kinv, near_far, cam2field, n_depth=64, depth=None):
    """Fused sample_cam_rays + cam_to_field: returns xyz_cam, dir_cam, deltas, depth, xyz_field, dir_field."""
    return _RaySamples.apply(hxy, Kinv, near_far, depth, cam2field[0], cam2field[1], n_depth)


class _Composite(Function):
    @staticmethod
    def forward(ctx, density, deltas, flow, vis, gdens, modes, *fields):
        M, N, D = density.shape[:3]
        R = M * N
        density, deltas = density.contiguous(), deltas.contiguous()
        fields = [f.contiguous() for f in fields]
        flow = flow.contiguous() if flow is not None else None
        vis = vis.contiguous() if vis is not None else None
        gdens = gdens.contiguous() if gdens is not None else None
        _f32(density, deltas, flow, vis, gdens, *fields)
        dev = density.device
        fl = _lib.FieldList()
        fl.n_fields = len(fields)
        sumC = 0
        for i, (f, m) in enumerate(zip(fields, modes)):
            fl.fields[i] = _lib.dp(f)
            fl.channels[i] = f.shape[-1]
            fl.modes[i] = m
            sumC += 1 if m == 2 else f.shape[-1]
        weights = torch.empty(M, N, D, device=dev)
        transmit = torch.empty(M, N, D, device=dev)
        mask = torch.empty(M, N, 1, device=dev)
        out = torch.empty(M, N, max(sumC, 1), device=dev)
        flow_out = torch.empty(M, N, 2, device=dev) if flow is not None else None
        vis_num = torch.empty(M, N, 1, device=dev) if vis is not None else None
        t_sum = torch.empty(M, N, 1, device=dev) if vis is not None else None
        gmask = torch.empty(M, N, 1, device=dev) if gdens is not None else None
        # algorithmic bytes: (2 + sum C) floats read per sample (+ flow 3, vis 1, gauss density 1), weights / transmittance written
        extra = (3 if flow is not None else 0) + (1 if vis is not None else 0) + (1 if gdens is not None else 0)
        chans = sum(f.shape[-1] for f in fields)
        ctx.chans = chans + extra
        with _lib.timed("k_composite_fwd", (0.0, 4.0 * R * D * (2 + chans + extra + 2))):
            _lib.check(_lib.lib().lab4d_composite_forward(
                _lib.ptr(density), _lib.ptr(deltas), fl, _lib.ptr(flow), _lib.ptr(vis), _lib.ptr(gdens), R, D, _lib.ptr(weights),
                _lib.ptr(transmit), _lib.ptr(mask), _lib.ptr(out), _lib.ptr(flow_out), _lib.ptr(vis_num), _lib.ptr(t_sum),
                _lib.ptr(gmask), _lib.stream()), "composite_forward")
        ctx.save_for_backward(density, deltas, flow, vis, gdens, *fields)
        ctx.modes = tuple(modes)
        ctx.dims = (M, N, D, sumC)
        ctx.mark_non_differentiable(weights, transmit)
        if t_sum is not None:
            ctx.mark_non_differentiable(t_sum)
        return mask, out, flow_out, vis_num, t_sum, gmask, weights, transmit

    @staticmethod
    @once_differentiable
    def backward(ctx, g_mask, g_out, g_flow_out, g_vis_num, g_t_sum, g_gmask, g_w, g_T):
        density, deltas, flow, vis, gdens, *fields = ctx.saved_tensors
        M, N, D, sumC = ctx.dims
        R = M * N

        def c(g):
            return g.contiguous() if g is not None else None

        g_mask, g_out, g_flow_out, g_vis_num, g_gmask = c(g_mask), c(g_out), c(g_flow_out), c(g_vis_num), c(g_gmask)
        fl = _lib.FieldList()
        gf = _lib.FieldGrads()
        fl.n_fields = gf.n_fields = len(fields)
        gfields = []
        for i, (f, m) in enumerate(zip(fields, ctx.modes)):
            fl.fields[i] = _lib.dp(f)
            fl.channels[i] = f.shape[-1]
            fl.modes[i] = m
            need = ctx.needs_input_grad[6 + i]
            g = torch.empty_like(f) if need else None
            gfields.append(g)
            gf.fields[i] = _lib.dp(g) if g is not None else None
        g_density = torch.empty_like(density)
        g_deltas = torch.empty_like(deltas)
        g_flow = torch.empty_like(flow) if (flow is not None and ctx.needs_input_grad[2]) else None
        g_vis = torch.empty_like(vis) if (vis is not None and ctx.needs_input_grad[3]) else None
        g_gd = torch.empty_like(gdens) if (gdens is not None and ctx.needs_input_grad[4]) else None
        with _lib.timed("k_composite_bwd", (0.0, 4.0 * R * D * 2 * (2 + ctx.chans))):  # every per-sample field read once, its gradient written once
            _lib.check(_lib.lib().lab4d_composite_backward(
                _lib.ptr(density), _lib.ptr(deltas), fl, _lib.ptr(flow), _lib.ptr(vis), _lib.ptr(gdens), R, D, _lib.ptr(g_mask),
                _lib.ptr(g_out), _lib.ptr(g_flow_out), _lib.ptr(g_vis_num), _lib.ptr(g_gmask), _lib.ptr(g_density),
                _lib.ptr(g_deltas), gf, _lib.ptr(g_flow), _lib.ptr(g_vis), _lib.ptr(g_gd), _lib.stream()), "composite_backward")
        return (g_density, g_deltas, g_flow, g_vis, g_gd, None, *gfields)


def _composite(field_dict, deltas):
    """Runs the fused compositor over a reference-style field_dict; returns the raw pieces."""
    keys, modes, fields = [], [], []
    for k, v in field_dict.items():
        if k in KEY_MEAN:
            keys.append(k); modes.append(2); fields.append(v)
        elif k in KEY_SKIP:
            continue
        else:
            keys.append(k); modes.append(1 if k in KEY_FREEZE else 0); fields.append(v)
    flow = field_dict.get("flow")
    vis = field_dict.get("vis")
    gd = field_dict.get("gauss_density")
    if len(fields) > 16:
        raise RuntimeError("render_pixel: more than 16 fields")
    res = _Composite.apply(field_dict["density"], deltas, flow, vis, gd, tuple(modes), *fields)
    return keys, modes, fields, res


class _Weights(Function):
    """compute_weights with a device-side adjoint (used outside the fused render_pixel path)."""

    @staticmethod
    def forward(ctx, density, deltas):
        res = _Composite.apply(density.detach(), deltas.detach(), None, None, None, ())
        w, T = res[6], res[7]
        ctx.save_for_backward(density, deltas, w, T)
        ctx.mark_non_differentiable(T)
        return w, T

    @staticmethod
    @once_differentiable
    def backward(ctx, gw, gT):
        density, deltas, w, T = ctx.saved_tensors
        # dL/dtau_d = gw_d T_d - sum_{i>d} gw_i w_i
        x = gw * w
        suffix = torch.flip(torch.cumsum(torch.flip(x, [-1]), -1), [-1]) - x
        gtau = (gw * T - suffix)[..., None]
        return gtau * deltas, gtau * density


def compute_weights(density, deltas):
    """render_utils.py:99-126 -> (weights, transmit), both (M,N,D)."""
    return _Weights.apply(density, deltas)


def integrate(field_dict, weights):
    """render_utils.integrate (render_utils.py:129-184) for caller-supplied weights (M,N,D).  The renderer itself never calls
    it -- render_pixel() below runs weights + integration + visibility loss as one kernel pair -- so this entry point of the
    public signature is plain device tensor algebra (one reduction per key), kept for callers that bring their own weights."""
    _lib.require_device(weights)
    skip = ("density", "vis", "flow", "eikonal", "xy_reproj", "xyz_reproj", "gauss_density")
    freeze = ("cyc_dist", "xyz_cam", "skin_entropy")
    out = {"mask": weights.sum(-1, keepdim=True)}
    wn = weights / (out["mask"] + 1e-6)
    for k, v in field_dict.items():
        if k in skip:
            continue
        out[k] = ((wn.detach() if k in freeze else wn).unsqueeze(-1) * v).sum(-2)
    if "flow" in field_dict:
        wf = weights * field_dict["flow"][..., 2]
        wf = wf / (wf.sum(-1, keepdim=True) + 1e-6)
        out["flow"] = (wf.unsqueeze(-1) * field_dict["flow"][..., :2]).sum(-2)
    if "normal" in field_dict:
        out["normal"] = F.normalize(out["normal"], 2, -1)
    dkeys = [k for k in out if "density_" in k]
    if dkeys:
        dsum = torch.cat([out[k] for k in dkeys], -1).sum(-1, keepdim=True) + 1e-6
        for k in dkeys:
            out[k.replace("density_", "mask_")] = out[k] / dsum
            del out[k]
    return out


def render_pixel(field_dict, deltas):
    """Same contract as render_utils.render_pixel (render_utils.py:59-96)."""
    keys, modes, fields, (mask, out, flow_out, vis_num, t_sum, gmask, w, T) = _composite(field_dict, deltas)
    rendered = {"mask": mask}
    co = 0
    for k, m, f in zip(keys, modes, fields):
        c = 1 if m == 2 else f.shape[-1]
        v = out[..., co:co + c]
        co += c
        rendered[k] = v[..., 0] if m == 2 else v  # means are (M,N) in the reference (render_utils.py:76,79)
    if flow_out is not None:
        rendered["flow"] = flow_out
    if "normal" in rendered:
        rendered["normal"] = F.normalize(rendered["normal"], 2, -1)
    dkeys = [k for k in rendered if "density_" in k]
    if dkeys:
        dsum = torch.cat([rendered[k] for k in dkeys], -1).sum(-1, keepdim=True) + 1e-6
        for k in dkeys:
            rendered[k.replace("density_", "mask_")] = rendered[k] / dsum
            del rendered[k]
    if vis_num is not None:
        M, N, D = field_dict["density"].shape[:3]
        rendered["vis"] = vis_num / (t_sum.sum() / (M * N * D)).detach()
    if gmask is not None:
        rendered["gauss_mask"] = gmask
    return rendered


def sample_pdf(bins, weights, N_importance, det=False, eps=1e-5, return_inds=False):
    """render_utils.sample_pdf (render_utils.py:187-233).  det=True: u = linspace(0, 1, N_importance) formed in the kernel.
    det=False (render_utils.py:212-213; unused in-tree: importance sampling is eval-only, nerf.py:721-727): u is drawn here
    with the reference's own call, sorted per ray for the kernel's one-pass sweep of the cdf, and the result is put back in
    draw order -- inverse-CDF sampling acts on each u independently."""
    bins, weights = bins.contiguous(), weights.contiguous()
    _f32(bins, weights)
    R, n_w = weights.shape
    if bins.shape != (R, n_w + 1):
        raise RuntimeError("sample_pdf: bins must be (R, n_w+1)")
    samples = torch.empty(R, N_importance, device=bins.device)
    inds = torch.empty(R, N_importance, dtype=torch.int64, device=bins.device)
    if det:
        _lib.check(_lib.lib().lab4d_sample_pdf(_lib.ptr(bins), _lib.ptr(weights), R, n_w, N_importance, float(eps), _lib.ptr(samples),
                                               _lib.ptr(inds), _lib.stream()), "sample_pdf")
    else:
        u = torch.rand(R, N_importance, device=bins.device)
        us, order = torch.sort(u, -1)
        us = us.contiguous()
        _lib.check(_lib.lib().lab4d_sample_pdf_u(_lib.ptr(bins), _lib.ptr(weights), _lib.ptr(us), R, n_w, N_importance, float(eps),
                                                 _lib.ptr(samples), _lib.ptr(inds), _lib.stream()), "sample_pdf_u")
        samples = torch.empty_like(samples).scatter_(1, order, samples)
        inds = torch.empty_like(inds).scatter_(1, order, inds)
    return (samples, inds) if return_inds else samples


def sort_depth(a, b):
    """sort(cat([a, b], -1), -1) for two per-ray depth lists (nerf.py:731)."""
    a, b = a.contiguous(), b.contiguous()
    _f32(a, b)
    R = a.shape[0]
    out = torch.empty(R, a.shape[1] + b.shape[1], device=a.device)
    _lib.check(_lib.lib().lab4d_sort_depth(_lib.ptr(a), a.shape[1], _lib.ptr(b), b.shape[1], R, _lib.ptr(out), _lib.stream()),
               "sort_depth")
    return out


# ---------------------------------------------------------------------------------------------------
# valid-sample compaction of the evaluation path (include/lab4d_hip.h section 2b; nerf.py:495-528, 769-819)
# ---------------------------------------------------------------------------------------------------
_lib.register("lab4d_valid_mask", [_lib.vp] * 4 + [ctypes.c_long, _lib.vp, _lib.vp])
_lib.register("lab4d_compact", [_lib.vp, ctypes.c_long, _lib.vp, _lib.vp, _lib.vp, _lib.vp])
_lib.register("lab4d_compact_work_ints", [ctypes.c_long])
_lib.register("lab4d_gather_rows", [_lib.vp] * 3 + [ctypes.c_long, _lib.ci, _lib.vp, _lib.vp])
_lib.register("lab4d_scatter_rows", [_lib.vp] * 3 + [ctypes.c_long, _lib.ci, _lib.vp, _lib.vp])
_lib.register("lab4d_frame_of", [_lib.vp] * 2 + [ctypes.c_long, _lib.ci, _lib.vp, _lib.vp])


@torch.no_grad()
def valid_mask(xyz, xyz_t, aabb, t_aabb=None):
    """uint8 (S): xyz strictly inside aabb (2,3) and (if t_aabb is given) xyz_t strictly inside t_aabb (2,3)."""
    x = xyz.reshape(-1, 3).contiguous()
    xt = xyz_t.reshape(-1, 3).contiguous() if t_aabb is not None else None
    aabb = aabb.contiguous().float()
    t_aabb = None if t_aabb is None else t_aabb.contiguous().float()
    _lib.require_device(x, xt, aabb, t_aabb)
    mask = torch.empty(x.shape[0], dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().lab4d_valid_mask(_lib.ptr(x), _lib.ptr(xt), _lib.ptr(aabb), _lib.ptr(t_aabb), x.shape[0], _lib.ptr(mask), _lib.stream()),
               "valid_mask")
    return mask


@torch.no_grad()
def compact(mask):
    """Stream compaction of a uint8 mask (S): (idx int32 (S): ascending indices of the set entries in its first `count` slots,
    count int32 device scalar).  Nothing is synchronised with the host."""
    mask = mask.contiguous()
    _lib.require_device(mask)
    S = mask.numel()
    idx = torch.empty(S, dtype=torch.int32, device=mask.device)
    count = torch.empty(1, dtype=torch.int32, device=mask.device)
    work = torch.empty(int(_lib.lib().lab4d_compact_work_ints(S)), dtype=torch.int32, device=mask.device)
    _lib.check(_lib.lib().lab4d_compact(_lib.ptr(mask), S, _lib.ptr(idx), _lib.ptr(count), _lib.ptr(work), _lib.stream()), "compact")
    return idx, count


@torch.no_grad()
def gather_rows(src, idx, count):
    """(S,C) -> (S,C): row j = src[idx[j]] for j < count, zeros after."""
    src = src.contiguous().float()
    out = torch.empty_like(src)
    _lib.check(_lib.lib().lab4d_gather_rows(_lib.ptr(src), _lib.ptr(idx), _lib.ptr(count), src.shape[0], src.shape[1], _lib.ptr(out), _lib.stream()),
               "gather_rows")
    return out


class _GatherRowsAD(torch.autograd.Function):
    """(S,C) -> (cap,C): row j = src[idx[j]] for j < count, zeros after; the adjoint scatters the row gradients back (rows nobody gathered: zero)."""

    @staticmethod
    def forward(ctx, src, idx, count, cap):
        src = src.contiguous().float()
        _lib.require_device(src, idx, count)
        out = torch.empty(int(cap), src.shape[1], device=src.device)
        _lib.check(_lib.lib().lab4d_gather_rows(_lib.ptr(src), _lib.ptr(idx), _lib.ptr(count), int(cap), src.shape[1], _lib.ptr(out), _lib.stream()), "gather_rows")
        ctx.save_for_backward(idx, count)
        ctx.n_rows = src.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        idx, count = ctx.saved_tensors
        g = g.contiguous().float()
        dst = torch.zeros(ctx.n_rows, g.shape[1], device=g.device)
        _lib.check(_lib.lib().lab4d_scatter_rows(_lib.ptr(g), _lib.ptr(idx), _lib.ptr(count), g.shape[0], g.shape[1], _lib.ptr(dst), _lib.stream()), "scatter_rows")
        return dst, None, None, None


class _ScatterRowsAD(torch.autograd.Function):
    """(cap,C) -> zeros (n_rows,C) with row idx[j] = src[j] for j < min(count, cap); the adjoint gathers."""

    @staticmethod
    def forward(ctx, src, idx, count, n_rows):
        src = src.contiguous().float()
        _lib.require_device(src, idx, count)
        dst = torch.zeros(int(n_rows), src.shape[1], device=src.device)
        _lib.check(_lib.lib().lab4d_scatter_rows(_lib.ptr(src), _lib.ptr(idx), _lib.ptr(count), src.shape[0], src.shape[1], _lib.ptr(dst), _lib.stream()), "scatter_rows")
        ctx.save_for_backward(idx, count)
        ctx.cap = src.shape[0]
        return dst

    @staticmethod
    def backward(ctx, g):
        idx, count = ctx.saved_tensors
        g = g.contiguous().float()
        out = torch.empty(ctx.cap, g.shape[1], device=g.device)
        _lib.check(_lib.lib().lab4d_gather_rows(_lib.ptr(g), _lib.ptr(idx), _lib.ptr(count), ctx.cap, g.shape[1], _lib.ptr(out), _lib.stream()), "gather_rows")
        return out, None, None, None


def gather_rows_ad(src, idx, count, cap):
    """Differentiable gather_rows into a buffer of `cap` rows (a static capacity: the count stays on the device)."""
    return _GatherRowsAD.apply(src, idx, count, cap)


def scatter_rows_ad(src, idx, count, n_rows):
    """Differentiable scatter_rows of the first min(count, rows of src) rows of src into zeros (n_rows, C)."""
    return _ScatterRowsAD.apply(src, idx, count, n_rows)


@torch.no_grad()
def scatter_rows(src, idx, count, n_rows):
    """zeros (n_rows,C) with row idx[j] = src[j] for j < count (query_nerf's scatter into zeros, nerf.py:812-816)."""
    src = src.contiguous().float()
    out = torch.zeros(n_rows, src.shape[1], device=src.device)
    _lib.check(_lib.lib().lab4d_scatter_rows(_lib.ptr(src), _lib.ptr(idx), _lib.ptr(count), src.shape[0], src.shape[1], _lib.ptr(out), _lib.stream()),
               "scatter_rows")
    return out


@torch.no_grad()
def frame_of(idx, count, spf):
    out = torch.empty_like(idx)
    _lib.check(_lib.lib().lab4d_frame_of(_lib.ptr(idx), _lib.ptr(count), idx.numel(), int(spf), _lib.ptr(out), _lib.stream()), "frame_of")
    return out
