"""The row arithmetic of the skeleton-FK kernels (lab4d_amd/csrc/fk_math.hpp), compiled for the CPU with g++
(tests/host_harness/fk_host.cpp) and held to the reference-generated fixture tests/golden/pose.pt and to the oracle.
CPU only: this checks the math the gfx950 kernels execute without needing a GPU; tests/test_gpu_zpose.py runs the kernels."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32 = ctypes.POINTER(ctypes.c_float)
I32 = ctypes.POINTER(ctypes.c_int)


@pytest.fixture(scope="module")
def host():
    out = os.path.join(ROOT, "tests", "host_harness", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "fk_host.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "lab4d_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_harness", "fk_host.cpp"), "-o", so])
    return ctypes.CDLL(so)


@pytest.fixture(scope="module")
def pose(golden_dir):
    return torch.load(os.path.join(golden_dir, "pose.pt"), weights_only=False)


def fp(a):
    return a.ctypes.data_as(F32)


def ip(a):
    return a.ctypes.data_as(I32)


def arr(t):
    return np.ascontiguousarray(t.detach().numpy(), dtype=np.float32)


def skel_arrays(edges, B):
    order = np.asarray([k - 1 for k in edges.keys()], dtype=np.int32)
    parent = np.full(B, -1, dtype=np.int32)
    for k, p in edges.items():
        parent[k - 1] = p - 1
    return order, parent


def relerr(a, b):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def test_fk_rows_match_the_reference(host, pose):
    fk, edges = pose["fk"], pose["skel"]["edges"]
    so3, local, shift = arr(fk["so3"]), arr(fk["local"]), arr(fk["shift"])
    R, B = so3.shape[:2]
    order, parent = skel_arrays(edges, B)
    qr, qd = np.empty((R, B, 4), np.float32), np.empty((R, B, 4), np.float32)
    host.fk_host_forward(fp(so3), fp(local), None, ip(order), ip(parent), R, B, 0, fp(qr), fp(qd))
    assert relerr(qr, fk["joints_dq"][0]) < 1e-5 and relerr(qd, fk["joints_dq"][1]) < 1e-5
    c = [arr(x) for x in fk["cot"]]
    g_so3, g_loc, g_sh = np.empty_like(so3), np.empty_like(local), np.empty((R, 3), np.float32)
    host.fk_host_backward(fp(so3), fp(local), None, ip(order), ip(parent), fp(c[0]), fp(c[1]), R, B, 0, fp(g_so3), fp(g_loc), fp(g_sh))
    assert relerr(g_so3, fk["g_joints"][0]) < 1e-4 and relerr(g_loc, fk["g_joints"][1]) < 1e-4
    # + shift_joints_to_bones_dq
    host.fk_host_forward(fp(so3), fp(local), fp(shift), ip(order), ip(parent), R, B, 1, fp(qr), fp(qd))
    assert relerr(qr, fk["bones_dq"][0]) < 1e-5 and relerr(qd, fk["bones_dq"][1]) < 1e-5
    host.fk_host_backward(fp(so3), fp(local), fp(shift), ip(order), ip(parent), fp(c[2]), fp(c[3]), R, B, 1, fp(g_so3), fp(g_loc), fp(g_sh))
    assert relerr(g_so3, fk["g_bones"][0]) < 1e-4 and relerr(g_loc, fk["g_bones"][1]) < 1e-4
    assert relerr(g_sh.sum(0), fk["g_bones"][2]) < 1e-4


def test_skel_rows_match_the_oracle(host, pose):
    """Fused bone lengths + FK + bones against oracle.rel_rest_joints / fk_se3 / shift_joints_to_bones_dq, with gradients wrt
    so3, the log bone lengths, logscale and shift; also an `edges` order in which a child is visited before its parent."""
    from oracle import pose_oracle as PO
    skel = pose["skel"]
    B = skel["rest_joints"].shape[0]
    g = torch.Generator().manual_seed(2)
    R = 7
    rest_local = PO.rest_joints_to_local(skel["rest_joints"], skel["edges"])
    keys = list(skel["edges"].keys())
    shuffled = dict((k, skel["edges"][k]) for k in [keys[i] for i in torch.randperm(len(keys), generator=g).tolist()])
    for edges in (skel["edges"], shuffled):
        so3 = (torch.randn(R, B, 3, generator=g) * 1.2).requires_grad_(True)
        ll = (torch.randn(R, B, generator=g) * 0.3).requires_grad_(True)
        ls = torch.tensor(-0.2, requires_grad=True)
        shift = torch.tensor([0.03, -0.01, 0.02], requires_grad=True)
        length = (ll + ls).exp()
        length = (length + length[:, skel["symm_idx"]]) / 2
        dq = PO.fk_se3(rest_local[None] * length[..., None], so3, edges)
        br, bd = PO.shift_joints_to_bones_dq(dq, edges, shift=shift)
        c0, c1 = torch.randn(R, B, 4, generator=g), torch.randn(R, B, 4, generator=g)
        ref = torch.autograd.grad((br * c0).sum() + (bd * c1).sum(), [so3, ll, ls, shift])
        order, parent = skel_arrays(edges, B)
        symm = np.asarray(skel["symm_idx"], dtype=np.int32)
        a_so3, a_ll, a_rest, a_sh = arr(so3), arr(ll), arr(rest_local), arr(shift)
        qr, qd = np.empty((R, B, 4), np.float32), np.empty((R, B, 4), np.float32)
        host.skel_host_forward(fp(a_so3), fp(a_ll), ctypes.c_float(float(ls.detach())), fp(a_rest), fp(a_sh), ip(order), ip(parent), ip(symm), R, B, fp(qr), fp(qd))
        assert relerr(qr, br.detach()) < 1e-5 and relerr(qd, bd.detach()) < 1e-5
        g_so3, g_ll, g_ls, g_sh = np.empty_like(a_so3), np.empty_like(a_ll), np.empty(R, np.float32), np.empty((R, 3), np.float32)
        host.skel_host_backward(fp(a_so3), fp(a_ll), ctypes.c_float(float(ls.detach())), fp(a_rest), fp(a_sh), ip(order), ip(parent), ip(symm), fp(arr(c0)),
                                fp(arr(c1)), R, B, fp(g_so3), fp(g_ll), fp(g_ls), fp(g_sh))
        assert relerr(g_so3, ref[0]) < 1e-4 and relerr(g_ll, ref[1]) < 1e-4
        assert relerr(g_ls.sum(), ref[2]) < 1e-4 and relerr(g_sh.sum(0), ref[3]) < 1e-4


def test_fk_rows_on_random_trees(host):
    """hypothesis: random skeletons (1..32 joints, random parents, random visiting order incl. child-before-parent), random
    angles incl. exact zeros: the row arithmetic equals the oracle's fk_se3 + shift_joints_to_bones_dq, forward and adjoint."""
    from hypothesis import given, settings
    from hypothesis import strategies as st
    from oracle import pose_oracle as PO

    @given(st.integers(0, 10_000), st.integers(1, 32), st.booleans(), st.booleans())
    @settings(max_examples=30, deadline=None)
    def run(seed, B, topo, bones):
        g = torch.Generator().manual_seed(seed)
        parents = [0] + [int(torch.randint(0, j + 1, (1,), generator=g)) for j in range(1, B)]  # joint j+1 hangs off 0..j (0 = root)
        keys = list(range(1, B + 1))
        if not topo:
            keys = [keys[i] for i in torch.randperm(B, generator=g).tolist()]
        edges = {k: parents[k - 1] for k in keys}
        R = 3
        so3 = torch.randn(R, B, 3, generator=g) * 1.3
        so3[0, ::3] = 0
        local = torch.randn(R, B, 3, generator=g) * 0.1
        shift = torch.randn(3, generator=g) * 0.05 if bones else None
        so3_t, local_t = so3.clone().requires_grad_(True), local.clone().requires_grad_(True)
        shift_t = shift.clone().requires_grad_(True) if bones else None
        dq = PO.fk_se3(local_t, so3_t, edges)
        if bones:
            dq = PO.shift_joints_to_bones_dq(dq, edges, shift=shift_t)
        c0, c1 = torch.randn(R, B, 4, generator=g), torch.randn(R, B, 4, generator=g)
        ref = torch.autograd.grad((dq[0] * c0).sum() + (dq[1] * c1).sum(), [so3_t, local_t] + ([shift_t] if bones else []))
        order, parent = skel_arrays(edges, B)
        a_so3, a_loc = arr(so3), arr(local)
        a_sh = arr(shift) if bones else None
        qr, qd = np.empty((R, B, 4), np.float32), np.empty((R, B, 4), np.float32)
        host.fk_host_forward(fp(a_so3), fp(a_loc), fp(a_sh) if bones else None, ip(order), ip(parent), R, B, int(bones), fp(qr), fp(qd))
        assert np.allclose(qr, dq[0].detach().numpy(), atol=2e-5) and np.allclose(qd, dq[1].detach().numpy(), atol=2e-5)
        g_so3, g_loc, g_sh = np.empty_like(a_so3), np.empty_like(a_loc), np.empty((R, 3), np.float32)
        host.fk_host_backward(fp(a_so3), fp(a_loc), fp(a_sh) if bones else None, ip(order), ip(parent), fp(arr(c0)), fp(arr(c1)), R, B, int(bones),
                              fp(g_so3), fp(g_loc), fp(g_sh))
        scale = max(1.0, float(ref[0].abs().max()))
        assert np.allclose(g_so3, ref[0].numpy(), atol=3e-4 * scale), float(np.abs(g_so3 - ref[0].numpy()).max())
        assert np.allclose(g_loc, ref[1].numpy(), atol=3e-4 * max(1.0, float(ref[1].abs().max())))
        if bones:
            assert np.allclose(g_sh.sum(0), ref[2].numpy(), atol=3e-4 * max(1.0, float(ref[2].abs().max())))

    run()


def test_fk_rows_equal_quaternion_composition(host, pose):
    """The reference's own known-answer test for this op (lab4d/tests/test_ops.py:181-267, test_fk): forward kinematics through
    4x4 matrices equals forward kinematics through quaternion-translation composition.  Here: the row arithmetic of the
    kernels (matrix chain + matrix_to_quaternion) against a quaternion-composition walk built from the oracle's quaternion ops,
    compared as SE(3) so the quaternion's sign convention drops out."""
    from oracle import lab4d_oracle as O
    from oracle import pose_oracle as PO
    from oracle import reg_oracle as RO
    skel = pose["skel"]
    edges, B = skel["edges"], skel["rest_joints"].shape[0]
    g = torch.Generator().manual_seed(8)
    R = 16
    so3 = torch.randn(R, B, 3, generator=g) * 1.4
    local = PO.rest_joints_to_local(skel["rest_joints"], edges)[None].expand(R, B, 3).contiguous()
    q_loc = PO.axis_angle_to_quaternion(so3)
    gq = [torch.tensor([1.0, 0, 0, 0]).expand(R, 4)] * B
    gt = [torch.zeros(R, 3)] * B
    for idx, par in edges.items():
        pq, pt = (gq[par - 1], gt[par - 1]) if par > 0 else (torch.tensor([1.0, 0, 0, 0]).expand(R, 4), torch.zeros(R, 3))
        gq[idx - 1] = O.quaternion_mul(pq, q_loc[:, idx - 1])
        gt[idx - 1] = O.quaternion_apply(pq, local[:, idx - 1]) + pt
    ref = RO.quaternion_translation_to_se3(torch.stack(gq, 1), torch.stack(gt, 1))
    order, parent = skel_arrays(edges, B)
    a_so3, a_loc = arr(so3), arr(local)
    qr, qd = np.empty((R, B, 4), np.float32), np.empty((R, B, 4), np.float32)
    host.fk_host_forward(fp(a_so3), fp(a_loc), None, ip(order), ip(parent), R, B, 0, fp(qr), fp(qd))
    qr_t, qd_t = torch.from_numpy(qr), torch.from_numpy(qd)
    t = 2 * O.quaternion_mul(qd_t, O.quaternion_conjugate(qr_t))[..., 1:]
    got = RO.quaternion_translation_to_se3(qr_t, t)
    assert torch.allclose(got, ref, atol=2e-5), float((got - ref).abs().max())
