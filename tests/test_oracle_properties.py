"""Size-independent properties the reference's algorithms imply, checked on the CPU oracle with
hypothesis-generated inputs.  CPU only; the GPU suite checks the same properties on the HIP path at benchmark scale."""
import math

import torch
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import lab4d_oracle as O

SET = dict(max_examples=25, deadline=None)


def _gen(seed):
    return torch.Generator().manual_seed(seed)


@given(st.integers(0, 10_000), st.integers(1, 4), st.integers(1, 9), st.integers(2, 33))
@settings(**SET)
def test_weights_and_final_transmittance_sum_to_one(seed, M, N, D):
    """compute_weights (render_utils.py:99-126): sum_D w + T_after_last = 1 for any density / deltas >= 0."""
    g = _gen(seed)
    dens = torch.rand(M, N, D, 1, generator=g) * 40
    deltas = torch.rand(M, N, D, 1, generator=g) * 0.1
    w, t = O.compute_weights(dens, deltas)
    tau = (dens * deltas)[..., 0]
    t_last = torch.exp(-tau.sum(-1))
    assert torch.allclose(w.sum(-1) + t_last, torch.ones(M, N), atol=2e-6)
    assert (w >= 0).all() and (t <= 1 + 1e-6).all()


@given(st.integers(0, 10_000), st.integers(1, 12), st.integers(3, 40), st.integers(1, 48))
@settings(**SET)
def test_sample_pdf_is_monotone_and_inside_the_bins(seed, R, nb, n_imp):
    """sample_pdf(det=True) (render_utils.py:187-233): samples are non-decreasing along the ray and stay in [bins.min, bins.max]."""
    g = _gen(seed)
    bins = torch.sort(torch.rand(R, nb + 1, generator=g), -1)[0]
    w = torch.rand(R, nb, generator=g)
    w[torch.rand(R, nb, generator=g) < 0.3] = 0  # empty bins
    s = O.sample_pdf(bins, w, n_imp)
    assert s.shape == (R, n_imp)
    assert (s[:, 1:] >= s[:, :-1] - 1e-7).all()
    assert (s >= bins[:, :1] - 1e-6).all() and (s <= bins[:, -1:] + 1e-6).all()


@given(st.integers(0, 10_000), st.integers(1, 50))
@settings(**SET)
def test_quaternion_algebra_identities(seed, n):
    g = _gen(seed)
    a, b = torch.randn(n, 4, generator=g), torch.randn(n, 4, generator=g)
    v = torch.randn(n, 3, generator=g)
    ab = O.quaternion_mul(a, b)
    assert torch.allclose(ab.norm(dim=-1), a.norm(dim=-1) * b.norm(dim=-1), rtol=1e-4, atol=1e-5)
    assert torch.equal(O.quaternion_conjugate(O.quaternion_conjugate(a)), a)
    # conj(ab) = conj(b) conj(a)
    assert torch.allclose(O.quaternion_conjugate(ab), O.quaternion_mul(O.quaternion_conjugate(b), O.quaternion_conjugate(a)), atol=1e-5)
    u = a / a.norm(dim=-1, keepdim=True)
    rv = O.quaternion_apply(u, v)
    assert torch.allclose(rv.norm(dim=-1), v.norm(dim=-1), rtol=1e-4, atol=1e-5)
    assert torch.allclose(O.quaternion_apply(O.quaternion_conjugate(u), rv), v, atol=1e-4)


@given(st.integers(0, 10_000), st.integers(1, 30))
@settings(**SET)
def test_dual_quaternion_inverse_and_composition(seed, n):
    g = _gen(seed)
    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    t = torch.randn(n, 3, generator=g)
    tq = torch.cat([torch.zeros(n, 1), t], -1)
    dq = (q, 0.5 * O.quaternion_mul(tq, q))  # quaternion_translation_to_dual_quaternion (quat_transform.py:290-297)
    p = torch.randn(n, 3, generator=g)
    moved = O.dual_quaternion_apply(dq, p)
    assert torch.allclose(moved, O.quaternion_apply(q, p) + t, atol=1e-4)
    back = O.dual_quaternion_apply(O.dual_quaternion_inverse(dq), moved)
    assert torch.allclose(back, p, atol=1e-4)
    ident = O.dual_quaternion_mul(dq, O.dual_quaternion_inverse(dq))
    assert torch.allclose(ident[0], torch.tensor([1.0, 0, 0, 0]).expand(n, 4), atol=1e-5)
    assert torch.allclose(ident[1], torch.zeros(n, 4), atol=1e-5)


@given(st.integers(0, 10_000), st.integers(0, 12))
@settings(**SET)
def test_posenc_layout_and_annealing_limits(seed, L):
    """PosEmbedding (embedding.py:69-125): channel layout [x, (freq, {sin,cos}, axis)]; alpha >= 1 opens every band."""
    g = _gen(seed)
    x = torch.randn(7, 3, generator=g)
    e = O.pos_embedding(x, L)
    assert e.shape == (7, 3 + 6 * L)
    assert torch.equal(e[:, :3], x)
    for f in range(L):
        assert torch.allclose(e[:, 3 + 6 * f:6 + 6 * f], torch.sin(x * 2.0 ** f), atol=1e-6)
        assert torch.allclose(e[:, 6 + 6 * f:9 + 6 * f], torch.cos(x * 2.0 ** f), atol=1e-6)
    if L:
        assert torch.allclose(O.pos_embedding(x, L, alpha=1.0), e, atol=1e-6)
        e0 = O.pos_embedding(x, L, alpha=0.0)  # window closed: only the raw coordinates survive
        assert torch.equal(e0[:, :3], x) and float(e0[:, 3:].abs().max()) < 1e-6


@given(st.integers(0, 10_000), st.integers(1, 3), st.integers(1, 6), st.integers(1, 9), st.integers(1, 9))
@settings(**SET)
def test_compose_fields_is_a_sorting_permutation(seed, M, N, Da, Db):
    """compose_fields (multifields.py:339-398): depth comes out sorted, every key is moved by the same permutation, keys
    missing from one field are zero-filled."""
    g = _gen(seed)
    fa = {"depth": torch.sort(torch.rand(M, N, Da, 1, generator=g), 2)[0], "rgb": torch.rand(M, N, Da, 3, generator=g) + 1}
    if Da == Db:  # the reference zero-fills a missing key with zeros_like(the OTHER field's tensor) (multifields.py:386-389),
        fa["only_a"] = torch.rand(M, N, Da, 1, generator=g) + 1  # which is only well-formed when both fields have the same D
    fb = {"depth": torch.sort(torch.rand(M, N, Db, 1, generator=g), 2)[0], "rgb": torch.rand(M, N, Db, 3, generator=g) + 1}
    out, deltas = O.compose_fields({"fg": fa, "bg": fb}, {"fg": torch.rand(M, N, Da, 1, generator=g), "bg": torch.rand(M, N, Db, 1, generator=g)})
    d = out["depth"]
    assert d.shape == (M, N, Da + Db, 1) and deltas.shape == (M, N, Da + Db, 1)
    assert (d[:, :, 1:] >= d[:, :, :-1]).all()
    # multiset of (depth, rgb) rows is preserved
    cat = torch.cat([torch.cat([fa["depth"], fa["rgb"]], -1), torch.cat([fb["depth"], fb["rgb"]], -1)], 2)
    got = torch.cat([out["depth"], out["rgb"]], -1)
    assert torch.equal(torch.sort(cat.sum(-1), 2)[0], torch.sort(got.sum(-1), 2)[0])
    if Da == Db:
        assert int((out["only_a"] == 0).sum()) == M * N * Db  # bg rows are zero-filled (values of fg are >= 1)


def test_mat3x3_inverse_is_an_inverse():
    g = _gen(5)
    m = torch.randn(64, 3, 3, generator=g) + 2 * torch.eye(3)
    inv = O.mat3x3_inv(m)
    assert torch.allclose(inv @ m, torch.eye(3).expand(64, 3, 3), atol=1e-4)
    assert torch.allclose(O.mat3x3_det(m), torch.linalg.det(m), rtol=1e-4, atol=1e-5)


def test_gauss_density_peaks_at_a_bone_centre():
    """compute_gauss_density (deformable.py:329-356): max_b exp(-|x-c_b|^2 / (2 * 0.01^2)) * ibeta equals ibeta at a centre."""
    from lab4d_amd import synthetic
    P = synthetic.make_weights(0)
    fr = synthetic.make_frames(1, 2, 64)
    _, centres = O.dual_quaternion_to_quaternion_translation((fr["rest_articulation"][0][:1], fr["rest_articulation"][1][:1]))
    x = centres[0][None, :, None, :]  # (1, B, 1, 3): exactly at the centres
    d = O.gauss_density(P, x, fr["rest_articulation"])
    assert torch.allclose(d, torch.full_like(d, math.exp(float(P["warp.logibeta"]))), rtol=1e-5)


def test_delta_skin_first_layer_is_affine_in_the_point():
    """The identity behind LAB4D_NET_SKIN_A (include/lab4d_mlp.h): SkinningField.forward (skinning.py:89-124, restated in O.skinning_field) feeds
    linear_1 the gaussian-scaled bone coordinates through PosEmbedding(3B, 0) = identity, and those are affine in the point per (frame, bone),
    so  linear_1([coords | t_embed | code]) == Wf[frame] [x; 1]  with  Wf = W1[:, :3B] aff[frame]  (+ the conditioning columns and the bias in the
    last column).  Checked in float64 on the oracle's own functions: the table is read off get_bone_coords by evaluating it at 0, e_x, e_y, e_z."""
    from lab4d_amd import synthetic
    M, N, D, B = 3, 4, 5, 25
    P = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in synthetic.make_weights(2).items()}
    fr = synthetic.add_codes(synthetic.make_frames(3, M, 64), synthetic.make_weights(2))
    art = tuple(t.double() for t in fr["t_articulation"])
    t_embed, code = fr["t_embed"].double(), fr["code_skin"].double()
    g = torch.Generator().manual_seed(5)
    xyz = torch.randn(M, N, D, 3, generator=g, dtype=torch.float64) * 0.1
    q = "warp.skinning_model.delta_field."
    W1, b1 = P[q + "linear_1.0.weight"], P[q + "linear_1.0.bias"]
    gauss = O.get_gauss(P)

    def coords(pts):  # (M, K, 3) points -> (M, K, 3B) gaussian-scaled bone coordinates
        b2o = tuple(a[:, None].expand((M, pts.shape[1]) + a.shape[1:]) for a in art)
        return (O.get_bone_coords(pts, b2o) / gauss.view(1, 1, -1, 3)).reshape(M, pts.shape[1], 3 * B)

    probe = torch.cat([torch.zeros(1, 3, dtype=torch.float64), torch.eye(3, dtype=torch.float64)])[None].expand(M, 4, 3)
    c = coords(probe)                                                            # (M, 4, 3B)
    aff = torch.cat([(c[:, 1:] - c[:, :1]).transpose(1, 2), c[:, 0, :, None]], -1)  # (M, 3B, 4): row = [linear part | offset]
    pf = torch.cat([t_embed, code], -1) @ W1[:, 3 * B:].t()                       # (M, 64): the per-frame conditioning columns
    tab = torch.einsum("fc,mcj->mfj", W1[:, :3 * B], aff)
    tab = torch.cat([tab[..., :3], tab[..., 3:] + (pf + b1)[..., None]], -1)     # what warping.skin_affine_table builds
    xh = torch.cat([xyz, torch.ones_like(xyz[..., :1])], -1)
    z_affine = torch.einsum("mfj,mndj->mndf", tab, xh)
    emb = coords(xyz.reshape(M, N * D, 3)).reshape(M, N, D, 3 * B)
    cond = torch.cat([t_embed, code], -1)[:, None, None].expand(M, N, D, 160)
    z_ref = torch.nn.functional.linear(torch.cat([emb, cond], -1), W1, b1)
    assert float((z_affine - z_ref).abs().max()) < 1e-10 * float(z_ref.abs().max())
    # and therefore the whole field: the remaining two layers on relu(z) reproduce O.skinning_field's delta
    h = torch.relu(z_affine)
    h = torch.relu(torch.nn.functional.linear(h, P[q + "linear_2.0.weight"], P[q + "linear_2.0.bias"]))
    delta = torch.relu(torch.nn.functional.linear(h, P[q + "linear_final.weight"], P[q + "linear_final.bias"])) * 0.1
    _, delta_ref = O.skinning_field(P, xyz, art, t_embed, code)
    assert float((delta - delta_ref).abs().max()) < 1e-10
