"""GPU parity: HIP kernels (through the C-ABI) vs the CPU oracle on identical seeded inputs.
fp32 tolerance: rtol 1e-4 (north_star), atol scaled to the tensor's magnitude."""
import os

import pytest
import torch

from oracle import lab4d_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, name="", rtol=1e-4, atol=None):
    a = a.detach().float().cpu()
    b = b.detach().float()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if atol is None:
        atol = 1e-5 * max(1.0, float(b.abs().max()))
    err = (a - b).abs()
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"{name}: max abs err {float(err.max()):.3e}, ref max {float(b.abs().max()):.3e}"


def gen(seed):
    return torch.Generator().manual_seed(seed)


def test_library_loaded_is_the_in_tree_hip_library():
    from lab4d_amd import _lib
    assert _lib.lib().lab4d_arch() == b"gfx950"
    assert _lib.lib().lab4d_build_flags() == b"", "kernel-experiment macros in the shipped library: %r" % _lib.lib().lab4d_build_flags()
    assert os.path.samefile(_lib.SO_PATH, os.path.join(os.path.dirname(_lib.__file__), "liblab4d_hip.so"))


def test_mfma_layout_probe():
    from lab4d_amd import _lib
    import ctypes
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_harness", "_build", "libmfma_layout.so")
    assert os.path.exists(so), "build the probe first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(so)
    lib.mfma_layout_probe.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]
    g = gen(0)
    for use_bf16, K in [(1, 16), (0, 2)]:
        A = torch.randn(32, K, generator=g)
        B = torch.randn(K, 32, generator=g)
        if use_bf16:
            A, B = A.bfloat16().float(), B.bfloat16().float()
        D = torch.zeros(32, 32, device=DEV)
        Ad, Bd = A.to(DEV), B.to(DEV)
        assert lib.mfma_layout_probe(_lib.ptr(Ad), _lib.ptr(Bd), _lib.ptr(D), use_bf16, _lib.stream()) == 0
        close(D, A @ B, f"mfma bf16={use_bf16}", rtol=1e-5)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float64, 1e-12), (torch.float16, 2e-3)])
@pytest.mark.parametrize("D1,D2", [(4, 4), (4, 3), (3, 4)])
def test_quaternion_mul_fwd_bwd_bwdbwd(dtype, tol, D1, D2):
    from lab4d_amd.quaternion import quaternion_mul
    g = gen(1)
    B = 1000
    a = torch.randn(B, D1, generator=g, dtype=torch.float64)
    b = torch.randn(B, D2, generator=g, dtype=torch.float64)
    go = torch.randn(B, 4, generator=g, dtype=torch.float64)
    v1 = torch.randn(B, D1, generator=g, dtype=torch.float64)
    v2 = torch.randn(B, D2, generator=g, dtype=torch.float64)

    def run(fn, a, b, go, v1, v2):
        a = a.clone().requires_grad_(True)
        b = b.clone().requires_grad_(True)
        go = go.clone().requires_grad_(True)
        out = fn(a, b)
        ga, gb = torch.autograd.grad(out, [a, b], go, create_graph=True)
        gg, gga, ggb = torch.autograd.grad([ga, gb], [go, a, b], [v1, v2])
        return out, ga, gb, gg, gga, ggb

    ref = run(O.quaternion_mul, a, b, go, v1, v2)
    dev = run(quaternion_mul, *[t.to(DEV, dtype) for t in (a, b, go, v1, v2)])
    for r, d, n in zip(ref, dev, ["out", "ga", "gb", "gg", "gga", "ggb"]):
        close(d, r, n, rtol=tol * 10, atol=tol * 10 * float(r.abs().max()))


def test_quaternion_conjugate_and_empty():
    from lab4d_amd.quaternion import quaternion_conjugate, quaternion_mul
    q = torch.randn(257, 4, generator=gen(2))
    close(quaternion_conjugate(q.to(DEV)), O.quaternion_conjugate(q), "conj", atol=0)
    e = torch.empty(0, 4, device=DEV)
    assert quaternion_mul(e, e).shape == (0, 4)


def test_mat3x3():
    from lab4d_amd.quaternion import mat3x3_inv, mat3x3_det, mat3x3_scale_adjoint
    g = gen(3)
    m = torch.randn(513, 3, 3, generator=g) + 2 * torch.eye(3)
    md = m.to(DEV).requires_grad_(True)
    mr = m.clone().requires_grad_(True)
    close(mat3x3_det(md.detach()), O.mat3x3_det(m), "det")
    close(mat3x3_scale_adjoint(md.detach(), mat3x3_det(md.detach())), O.mat3x3_inv(m), "adj", rtol=1e-3)
    inv = mat3x3_inv(md)
    close(inv, torch.linalg.inv(m), "inv", rtol=1e-3)
    w = torch.randn(513, 3, 3, generator=g)
    (gd,) = torch.autograd.grad(inv, md, w.to(DEV))
    (gr,) = torch.autograd.grad(torch.linalg.inv(mr), mr, w)
    close(gd, gr, "inv bwd", rtol=1e-3)


@pytest.mark.parametrize("M,N,D,with_depth", [(2, 37, 64, False), (3, 5, 7, True), (1, 300, 128, False)])
def test_ray_samples(M, N, D, with_depth):
    from lab4d_amd import render_utils as RU
    from lab4d_amd import synthetic
    g = gen(4)
    fr = synthetic.make_frames(5, M + (M % 2), 64)
    Kinv = fr["Kinv"][:M].clone()
    nf = fr["near_far"][:M].clone()
    q, t = fr["field2cam"][0][:M], fr["field2cam"][1][:M]
    hxy = torch.cat([torch.rand(M, N, 2, generator=g) * 64, torch.ones(M, N, 1)], -1)
    depth = None
    if with_depth:
        depth = torch.sort(torch.rand(M, N, D, 1, generator=g) * 0.4 + 0.4, 2)[0]
    wts = [torch.randn(M, N, D, c, generator=g) for c in (3, 3, 1, 3, 3)]

    def ref():
        K = Kinv.clone().requires_grad_(True)
        qq, tt = q.clone().requires_grad_(True), t.clone().requires_grad_(True)
        xyz, dr, dl, dp = O.sample_cam_rays(hxy, K, nf, n_depth=D, depth=depth)
        xf, df = O.cam_to_field(xyz, dr, (qq, tt))
        loss = sum((a * w).sum() for a, w in zip([xyz, dr, dl, xf, df], wts))
        return (xyz, dr, dl, dp, xf, df), torch.autograd.grad(loss, [K, qq, tt])

    def dev():
        K = Kinv.to(DEV).requires_grad_(True)
        qq, tt = q.to(DEV).requires_grad_(True), t.to(DEV).requires_grad_(True)
        c2f = O.quaternion_translation_inverse(qq, tt)  # small per-frame torch ops on device
        outs = RU.ray_samples(hxy.to(DEV), K, nf.to(DEV), c2f, n_depth=D, depth=None if depth is None else depth.to(DEV))
        xyz, dr, dl, dp, xf, df = outs
        loss = sum((a * w.to(DEV)).sum() for a, w in zip([xyz, dr, dl, xf, df], wts))
        return outs, torch.autograd.grad(loss, [K, qq, tt])

    (ro, rg), (do, dg) = ref(), dev()
    for a, b, n in zip(do, ro, ["xyz_cam", "dir_cam", "deltas", "depth", "xyz_field", "dir_field"]):
        close(a, b, n)
    for a, b, n in zip(dg, rg, ["gKinv", "gq", "gt"]):
        close(a, b, n, rtol=2e-4, atol=2e-4 * float(b.abs().max()))
    # the unfused mirror of sample_cam_rays
    xyz, dr, dl, dp = RU.sample_cam_rays(hxy.to(DEV), Kinv.to(DEV), nf.to(DEV), n_depth=D, depth=None if depth is None else depth.to(DEV))
    close(xyz, ro[0], "sample_cam_rays.xyz")


def _field_dict(g, M, N, D, train=True):
    fd = {
        "density": torch.rand(M, N, D, 1, generator=g) * 600.0 / D,
        "density_fg": None,
        "rgb": torch.rand(M, N, D, 3, generator=g),
        "vis": torch.randn(M, N, D, 1, generator=g) * 3,
        "xyz": torch.randn(M, N, D, 3, generator=g),
        "xyz_cam": torch.randn(M, N, D, 3, generator=g),
        "depth": torch.rand(M, N, D, 1, generator=g),
        "eikonal": torch.rand(M, N, D, 1, generator=g),
        "gauss_density": torch.rand(M, N, D, 1, generator=g) * 200.0 / D,
    }
    if train:
        fd.update({
            "flow": torch.cat([torch.randn(M, N, D, 2, generator=g), (torch.rand(M, N, D, 1, generator=g) > 0.3).float()], -1),
            "cyc_dist": torch.rand(M, N, D, 1, generator=g),
            "skin_entropy": torch.rand(M, N, D, 1, generator=g),
            "delta_skin": torch.rand(M, N, D, 1, generator=g),
            "feature": torch.randn(M, N, D, 16, generator=g),
        })
    else:
        fd["normal"] = torch.randn(M, N, D, 3, generator=g)
    fd["density_fg"] = fd["density"]
    return fd


@pytest.mark.parametrize("M,N,D,train", [(2, 33, 64, True), (1, 50, 128, True), (2, 9, 16, False), (1, 3, 200, True)])
def test_render_pixel_forward_backward(M, N, D, train):
    from lab4d_amd import render_utils as RU
    g = gen(6)
    fd = _field_dict(g, M, N, D, train)
    deltas = torch.rand(M, N, D, 1, generator=g) * 0.02

    def run(mod, dev):
        leaves = {}
        f = {}
        for k, v in fd.items():
            if k == "density_fg":
                continue
            t = v.to(dev).clone().requires_grad_(k not in ("flow",))
            leaves[k] = t
            f[k] = t
        f["density_fg"] = f["density"]
        dl = deltas.to(dev).clone().requires_grad_(True)
        leaves["deltas"] = dl
        out = mod.render_pixel(f, dl)
        gg = torch.Generator().manual_seed(99)
        loss = 0
        for k in sorted(out.keys()):
            loss = loss + (out[k] * torch.randn(out[k].shape, generator=gg).to(dev)).sum()
        names = sorted(k for k in leaves if leaves[k].requires_grad)
        grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
        return out, dict(zip(names, grads))

    ro, rg = run(O, "cpu")
    do, dg = run(RU, DEV)
    assert set(ro.keys()) == set(do.keys())
    for k in ro:
        close(do[k], ro[k], "rendered." + k, rtol=2e-4)
    for k in rg:
        if rg[k] is None:
            assert dg[k] is None or float(dg[k].abs().max()) == 0, k
            continue
        close(dg[k], rg[k], "grad." + k, rtol=5e-4, atol=5e-5 * float(rg[k].abs().max()) + 1e-9)


def test_compute_weights_partition_of_unity_at_full_size():
    """Size-independent property at BASELINE scale: sum_d w + T_last == 1 for every ray."""
    from lab4d_amd import render_utils as RU
    g = torch.Generator(device=DEV).manual_seed(0)
    M, N, D = 1, 512 * 64, 128
    dens = torch.rand(M, N, D, 1, device=DEV, generator=g) * 50
    deltas = torch.rand(M, N, D, 1, device=DEV, generator=g) * 0.01
    w, T = RU.compute_weights(dens, deltas)
    err = (w.sum(-1) + T[..., -1] - 1).abs().max()
    assert float(err) < 1e-5


def test_sample_pdf_indices_and_samples(golden_dir):
    from lab4d_amd import render_utils as RU
    ops = torch.load(os.path.join(golden_dir, "ops.pt"), weights_only=False)
    bins, wts, s_ref, inds_ref = ops["sample_pdf"]  # produced by the reference itself
    s, inds = RU.sample_pdf(bins.to(DEV), wts.to(DEV), 16, det=True, return_inds=True)
    _check_inds(inds.cpu(), s.cpu(), bins, wts, 16, "reference_golden")
    close(s, s_ref, "samples", rtol=1e-4)
    # larger random cases against the oracle: rows with all-positive weights, and rows with a run of 30 zero-weight bins -- there
    # the cdf climbs by eps / sum = 3e-7 per entry, i.e. 30 consecutive entries sit within a few ulp of each other and of any
    # query u that lands among them, which is where (and only where) the last-ulp difference between torch's vectorised fp32
    # row sum and the kernel's sequential sum can move an index by one bin.  The two populations are measured separately.
    g = gen(8)
    R, nw, ni = 5000, 62, 64
    bins = torch.sort(torch.rand(R, nw + 1, generator=g), -1)[0]
    wts = torch.rand(R, nw, generator=g) + 0.01  # keep pdf away from the eps=1e-5 switch (a discontinuity of the reference)
    flat = torch.zeros(R, dtype=torch.bool)
    flat[::7] = True
    wts[flat, 10:40] = 0
    s, inds = RU.sample_pdf(bins.to(DEV), wts.to(DEV), ni, det=True, return_inds=True)
    _check_inds(inds.cpu()[~flat], s.cpu()[~flat], bins[~flat], wts[~flat], ni, "random_positive_weights")
    _check_inds(inds.cpu()[flat], s.cpu()[flat], bins[flat], wts[flat], ni, "random_zero_weight_runs")
    s_o, inds_o = O.sample_pdf(bins, wts, ni, return_inds=True)
    same = (inds.cpu() == inds_o)  # at an fp tie inside a zero-weight bin the sample legitimately jumps a bin
    close(torch.where(same, s.cpu(), s_o), s_o, "samples vs oracle", rtol=1e-4, atol=1e-5)
    assert bool((s.cpu()[:, 1:] >= s.cpu()[:, :-1] - 1e-6).all()), "det=True samples must be monotone"


# Bounds on searchsorted indices that differ from torch's, per case: (interior queries, the u = 1 end point).  Measured on MI355X
# (profiles/r02_parity_sample_pdf_*.json).  The end point is special: the last cdf entry is sum(pdf) = 1 up to rounding, and
# whether it is <= 1.0f decides between index n_w and n_w + 1.  That rounding depends on the order in which the row
# normaliser sum(w + eps) is reduced -- torch's CPU kernel (vectorised, lane count depends on the AVX level of the build),
# torch's GPU reduction and a sequential sum give normalisers that differ in the last ulp, so the reference does not agree
# with ITSELF across its own platforms there; both choices interpolate to the same sample (bins[n_w]).  Interior queries can
# differ only where a cdf entry is within 3 ulp of u; none does on these cases: the interior indices are bit-exact.
SAMPLE_PDF_MISMATCH_MAX = {"reference_golden": (0, 0), "random_positive_weights": (0, 0), "random_zero_weight_runs": (0, 0)}  # round 3: the normaliser is summed in ATen's own order (csrc/sample_pdf_math.hpp) -- every index incl. the end point is bit-exact


def _check_inds(inds, samples, bins, wts, n_imp, tag):
    """Indices vs torch.searchsorted(right=True) on torch's own cdf (`reference_golden`: the reference's stored output)."""
    from parity_report import report
    s_ref, inds_ref = O.sample_pdf(bins, wts, n_imp, return_inds=True)
    mism = inds != inds_ref
    n_end = int(mism[:, -1].sum())
    n_int = int(mism[:, :-1].sum())
    report("sample_pdf_" + tag, {"interior_mismatch_count": float(n_int), "interior_total": float(mism[:, :-1].numel()),
                                 "endpoint_mismatch_count": float(n_end), "rows": float(len(inds))})
    if mism.any():
        w = wts + 1e-5
        cdf = torch.cat([torch.zeros(len(w), 1), torch.cumsum(w / w.sum(-1, keepdim=True), -1)], -1)
        u = torch.linspace(0, 1, n_imp).expand(len(w), n_imp)
        rows, cols = mism.nonzero(as_tuple=True)
        for r, c in zip(rows.tolist(), cols.tolist()):
            j = min(int(inds[r, c]), int(inds_ref[r, c]))
            assert abs(int(inds[r, c]) - int(inds_ref[r, c])) == 1
            assert abs(float(cdf[r, j]) - float(u[r, c])) <= 3 * 1.2e-7 * max(1.0, float(u[r, c])), (r, c)
    max_int, max_end_frac = SAMPLE_PDF_MISMATCH_MAX[tag]
    assert n_int <= max_int and n_end <= max_end_frac * len(inds), (tag, n_int, n_end, len(inds))


def test_sample_pdf_random_draws_and_stratified_rays():
    """The two branches of the public signatures no in-tree caller enables: sample_pdf(det=False) and
    sample_cam_rays(perturb=True).  The device draws its own torch.rand, so the checks are on what the branch must
    satisfy: det=False samples = the inverse-CDF map of SOME u in [0,1) per element -- recovered through the oracle with the
    device's draw replayed; perturb=True depths stay inside their strata and rays / deltas follow from them."""
    from lab4d_amd import render_utils as RU
    g = gen(21)
    R, nw, ni = 300, 30, 16
    bins = torch.sort(torch.rand(R, nw + 1, generator=g), -1)[0]
    wts = torch.rand(R, nw, generator=g) + 0.05
    torch.manual_seed(5)
    s, inds = RU.sample_pdf(bins.to(DEV), wts.to(DEV), ni, det=False, return_inds=True)
    torch.manual_seed(5)
    u = torch.rand(R, ni, device=DEV).cpu()  # the draw sample_pdf made
    w = wts + 1e-5
    cdf = torch.cat([torch.zeros(R, 1), torch.cumsum(w / w.sum(-1, keepdim=True), -1)], -1)
    inds_ref = torch.searchsorted(cdf, u.contiguous(), right=True)
    assert float((inds.cpu() != inds_ref).float().mean()) < 2e-3
    below, above = (inds_ref - 1).clamp_min(0), inds_ref.clamp_max(nw)
    c0, c1, b0, b1 = cdf.gather(1, below), cdf.gather(1, above), bins.gather(1, below), bins.gather(1, above)
    den = c1 - c0
    den[den < 1e-5] = 1
    s_ref = b0 + (u - c0) / den * (b1 - b0)
    ok = inds.cpu() == inds_ref
    assert torch.allclose(s.cpu()[ok], s_ref[ok], rtol=1e-4, atol=1e-5)
    assert not bool((s.cpu()[:, 1:] >= s.cpu()[:, :-1]).all()), "det=False samples come back in draw order, not sorted"
    # stratified rays
    M, N, D = 2, 50, 16
    hxy = torch.cat([torch.rand(M, N, 2, generator=g) * 64, torch.ones(M, N, 1)], -1).to(DEV)
    Kinv = torch.linalg.inv(torch.tensor([[64.0, 0, 32], [0, 64, 32], [0, 0, 1]]))[None].repeat(M, 1, 1).to(DEV)
    nf = torch.tensor([[0.4, 0.8], [0.5, 0.9]], device=DEV)
    xyz0, dir0, del0, dep0 = RU.sample_cam_rays(hxy, Kinv, nf, n_depth=D)
    xyz, dirs, deltas, depth = RU.sample_cam_rays(hxy, Kinv, nf, n_depth=D, perturb=True)
    mid = 0.5 * (dep0[:, :, :-1] + dep0[:, :, 1:])
    lower, upper = torch.cat([dep0[:, :, :1], mid], 2), torch.cat([mid, dep0[:, :, -1:]], 2)
    assert bool(((depth >= lower - 1e-6) & (depth <= upper + 1e-6)).all()) and float((depth - dep0).abs().max()) > 1e-3
    assert torch.allclose(dirs, dir0) and torch.allclose(xyz, xyz0 / dep0 * depth, rtol=1e-5, atol=1e-6)
    dn = (hxy @ Kinv.transpose(1, 2)).norm(dim=-1)[:, :, None, None]
    ref_d = torch.cat([depth[:, :, 1:] - depth[:, :, :-1], depth[:, :, -1:] - depth[:, :, -2:-1]], 2) * dn
    assert torch.allclose(deltas, ref_d, rtol=1e-5, atol=1e-7)


def test_integrate_with_caller_weights_equals_render_pixel():
    """render_utils.integrate(field_dict, weights) (public signature; the renderer itself uses the fused render_pixel): with the
    weights compute_weights returns it must reproduce render_pixel's integrated channels, and the oracle's integrate."""
    from lab4d_amd import render_utils as RU
    g = gen(22)
    M, N, D = 2, 9, 12
    fd = {"density": torch.rand(M, N, D, 1, generator=g) * 30, "rgb": torch.rand(M, N, D, 3, generator=g), "xyz": torch.randn(M, N, D, 3, generator=g),
          "cyc_dist": torch.rand(M, N, D, 1, generator=g), "flow": torch.cat([torch.randn(M, N, D, 2, generator=g), (torch.rand(M, N, D, 1, generator=g) > 0.3).float()], -1),
          "normal": torch.randn(M, N, D, 3, generator=g)}
    fd["density_fg"] = fd["density"]
    deltas = torch.rand(M, N, D, 1, generator=g) * 0.02
    fdd, dd = synthetic_to(fd), deltas.to(DEV)
    w, _ = RU.compute_weights(fdd["density"], dd)
    out = RU.integrate(fdd, w)
    ref = O.integrate(fd, O.compute_weights(fd["density"], deltas)[0])
    assert set(out.keys()) == set(ref.keys())
    for k, v in ref.items():
        close(out[k], v, "integrate." + k, rtol=1e-4)
    rp = RU.render_pixel(fdd, dd)
    for k in ("mask", "rgb", "xyz", "cyc_dist", "flow", "normal", "mask_fg"):
        close(out[k], rp[k].cpu(), "integrate vs render_pixel." + k, rtol=1e-4)


def synthetic_to(d):
    return {k: v.to(DEV) for k, v in d.items()}


def test_sort_depth_matches_torch_sort():
    from lab4d_amd import render_utils as RU
    g = gen(9)
    a = torch.sort(torch.rand(3000, 32, generator=g), -1)[0]
    b = torch.sort(torch.rand(3000, 32, generator=g), -1)[0]
    b[:, 5] = b[:, 6] + 1e-7  # a local inversion, as rounding in sample_pdf can produce
    out = RU.sort_depth(a.to(DEV), b.to(DEV)).cpu()
    assert torch.equal(out, torch.sort(torch.cat([a, b], -1), -1)[0])


def test_compose_fields_matches_oracle_and_reference_golden(golden_dir):
    """MultiFields.compose_fields on the device vs the oracle (itself pinned to the reference's output in ops.pt)."""
    import os
    from lab4d_amd import multifields
    from oracle import lab4d_oracle as O
    ops = torch.load(os.path.join(golden_dir, "ops.pt"), weights_only=False)
    fdA, fdB, dA, dB, comp, dcomp = ops["compose_fields"]
    dev = lambda d: {k: v.to(DEV) for k, v in d.items()}
    out, deltas = multifields.compose_fields({"fg": dev(fdA), "bg": dev(fdB)}, {"fg": dA.to(DEV), "bg": dB.to(DEV)})
    for k, v in comp.items():
        assert torch.equal(out[k].cpu(), v), k  # pure data movement: bit-exact
    assert torch.equal(deltas.cpu(), dcomp)
    # larger random case incl. ties and gradients
    g = torch.Generator().manual_seed(3)
    M, N, Da, Db = 2, 37, 64, 64
    mk = lambda D, c: torch.rand(M, N, D, c, generator=g)
    fa = {"depth": torch.sort(mk(Da, 1), 2)[0], "rgb": mk(Da, 3), "density": mk(Da, 1), "cyc_dist": mk(Da, 1)}
    fb = {"depth": torch.sort(mk(Db, 1), 2)[0], "rgb": mk(Db, 3), "density": mk(Db, 1)}
    fb["depth"][:, :, :5] = fa["depth"][:, :, 10:15]  # exact ties between the fields
    fb["depth"] = torch.sort(fb["depth"], 2)[0]
    da, db = mk(Da, 1), mk(Db, 1)

    def run(mod, to):
        A = {k: to(v).clone().requires_grad_(k != "depth") for k, v in fa.items()}
        B = {k: to(v).clone().requires_grad_(k != "depth") for k, v in fb.items()}
        o, d = mod.compose_fields({"fg": A, "bg": B}, {"fg": to(da), "bg": to(db)})
        w = {k: torch.rand(v.shape, generator=torch.Generator().manual_seed(len(k))) for k, v in o.items()}
        loss = sum((o[k] * to(w[k])).sum() for k in o if k != "depth")
        names = [("A", k) for k in A if k != "depth"] + [("B", k) for k in B if k != "depth"]
        gs = torch.autograd.grad(loss, [A[k] for n, k in names if n == "A"] + [B[k] for n, k in names if n == "B"])
        return o, d, dict(zip(names, gs))

    ro, rd, rg = run(O, lambda t: t)
    do, dd, dg = run(multifields, lambda t: t.to(DEV))
    for k in ro:
        assert torch.equal(do[k].cpu(), ro[k]), k
    assert torch.equal(dd.cpu(), rd)
    for k in rg:
        assert torch.equal(dg[k].cpu(), rg[k]), k


@pytest.mark.parametrize("S,p", [(1, 1.0), (63, 0.5), (1024, 0.0), (1025, 1.0), (65573, 0.176), (4 * 1024 * 1024 + 17, 0.3)])
def test_stream_compaction_matches_nonzero(S, p):
    """lab4d_compact (wave ballot + popcount, block counts, one-block scan, wave-prefix write): the index list equals
    torch.nonzero of the mask -- ascending, bit-exact -- and the count stays on the device; gather / scatter / frame_of honour it."""
    from lab4d_amd import render_utils as RU
    g = gen(S % 1000 + 5)
    mask = (torch.rand(S, generator=g) < p).to(torch.uint8)
    idx, count = RU.compact(mask.to(DEV))
    ref = torch.nonzero(mask).flatten().int()
    assert count.dtype == torch.int32 and int(count) == ref.numel()
    assert torch.equal(idx.cpu()[: ref.numel()], ref)
    src = torch.randn(S, 3, generator=g)
    gat = RU.gather_rows(src.to(DEV), idx, count).cpu()
    assert torch.equal(gat[: ref.numel()], src[ref.long()]) and float(gat[ref.numel():].abs().sum()) == 0.0
    sc = RU.scatter_rows(gat.to(DEV), idx, count, S).cpu()
    assert torch.equal(sc, src * mask[:, None].float())
    fr = RU.frame_of(idx, count, 7).cpu()
    assert torch.equal(fr[: ref.numel()], ref // 7) and int(fr[ref.numel():].abs().sum()) == 0


def test_valid_mask_is_bit_exact_at_the_box_faces():
    """check_inside_aabb uses strict inequalities (geom_utils.py:506-517): points exactly on a face are outside."""
    from lab4d_amd import render_utils as RU
    g = gen(31)
    aabb = torch.tensor([[-0.1, -0.2, -0.3], [0.2, 0.1, 0.4]])
    t_aabb = torch.tensor([[-0.5, -0.5, -0.5], [0.0, 0.5, 0.5]])
    x = torch.rand(5000, 3, generator=g) - 0.5
    xt = torch.rand(5000, 3, generator=g) * 1.2 - 0.6
    x[:6] = torch.tensor([[-0.1, 0, 0], [0.2, 0, 0], [0, -0.2, 0], [0, 0.1, 0], [0, 0, -0.3], [0, 0, 0.4]])  # on the six faces
    x[6] = torch.tensor([0.19999999, 0.0999999, 0.39999998])
    xt[7] = torch.tensor([0.0, 0.0, 0.0])  # on a face of the second box
    x[7] = 0.0
    ins = lambda p, b: ((p > b[:1]) & (p < b[1:])).all(-1)
    m = RU.valid_mask(x.to(DEV), xt.to(DEV), aabb.to(DEV), t_aabb.to(DEV)).cpu().bool()
    assert torch.equal(m, ins(x, aabb) & ins(xt, t_aabb))
    assert not m[:6].any() and not m[7]
    m1 = RU.valid_mask(x.to(DEV), None, aabb.to(DEV)).cpu().bool()
    assert torch.equal(m1, ins(x, aabb))


def test_compacted_chain_equals_the_chain_on_gathered_samples():
    """run_chain_compacted (device-side sample count, per-sample frame index) == run_chain on the same samples in frame order."""
    from lab4d_amd import mlp, render_utils as RU, synthetic
    P = synthetic.to_device(synthetic.make_weights(9), DEV)
    fr = synthetic.add_codes(synthetic.to_device(synthetic.make_frames(10, 2, 64), DEV), P)
    g = gen(33)
    M, spf = 2, 700
    x = ((torch.rand(M * spf, 3, generator=g) - 0.5) * 0.3).to(DEV)
    mask = (torch.rand(M * spf, generator=g) < 0.3).to(torch.uint8).to(DEV)
    idx, count = RU.compact(mask)
    K = int(count)
    x_k = RU.gather_rows(x, idx, count)
    frame_k = RU.frame_of(idx, count, spf)
    for prec in (mlp.PREC_F32, mlp.PREC_BF16):
        with torch.no_grad():
            full = mlp.run_chain(mlp.NET_VIS, prec, P, x, spf, conds={0: fr["code_vis"]})
            comp = mlp.run_chain_compacted(mlp.NET_VIS, prec, P, x_k, frame_k, count, conds={0: fr["code_vis"]})
            assert torch.allclose(comp[:K], full[idx[:K].long()], rtol=1e-5, atol=1e-6)
            sdf_f, feat_f = mlp.run_chain(mlp.NET_FG_BASE, prec, P, x, spf, conds={0: fr["code_base"], 4: fr["code_base"]}, export_layer=8)
            sdf_c, feat_c = mlp.run_chain_compacted(mlp.NET_FG_BASE, prec, P, x_k, frame_k, count, conds={0: fr["code_base"], 4: fr["code_base"]},
                                                    export_layer=8)
            assert torch.allclose(sdf_c[:K], sdf_f[idx[:K].long()], rtol=1e-5, atol=1e-6)
            rgb_f = mlp.run_chain(mlp.NET_FG_COLOR, prec, P, x, spf, conds={0: fr["code_color"], 3: fr["appr_code"]}, ext=feat_f)
            rgb_c = mlp.run_chain_compacted(mlp.NET_FG_COLOR, prec, P, x_k, frame_k, count, conds={0: fr["code_color"], 3: fr["appr_code"]}, ext=feat_c)
            assert torch.allclose(rgb_c[:K], rgb_f[idx[:K].long()], rtol=1e-5, atol=1e-6)
    # an empty selection: nothing is processed, nothing is written
    idx0, count0 = RU.compact(torch.zeros_like(mask))
    out = mlp.run_chain_compacted(mlp.NET_VIS, mlp.PREC_BF16, P, x_k, RU.frame_of(idx0, count0, spf), count0, conds={0: fr["code_vis"]})
    assert int(count0) == 0 and out.shape == (M * spf, 1)


def test_compose_order_with_nan_depths_is_still_a_permutation():
    """A NaN depth (e.g. parameters gone NaN upstream) must come out as NaN values, never as a wild gather index: NaNs sort last,
    like torch.sort, and `order` stays a permutation of every ray's samples."""
    from lab4d_amd import multifields
    g = torch.Generator().manual_seed(9)
    M, N, Da, Db = 2, 33, 64, 64
    a, b = torch.rand(M, N, Da, 1, generator=g), torch.rand(M, N, Db, 1, generator=g)
    a[0, 3, 5:9] = float("nan")
    b[1, 7, :] = float("nan")
    a[1, 8, :] = float("nan")
    b[1, 8, :] = float("nan")
    order, pos = multifields.compose_order(a.to(DEV), b.to(DEV))
    order, pos = order.cpu().long(), pos.cpu().long()
    R, Dt = M * N, Da + Db
    assert torch.equal(torch.sort(order, 1)[0], torch.arange(Dt).expand(R, Dt))
    assert torch.equal(torch.gather(pos, 1, order), torch.arange(Dt).expand(R, Dt))
    cat = torch.cat([a, b], 2).reshape(R, Dt)
    ref = torch.sort(cat, stable=True, dim=1)[1]
    assert torch.equal(order, ref)
