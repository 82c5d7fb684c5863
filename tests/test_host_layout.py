"""Host-side weight-layout logic of lab4d_amd/mlp.py on CPU (no launches): the column maps that turn reference-layout
(checkpoint compatible) weights into the kernels' input order must consume every reference column exactly once -- embedding
block through the slot permutation, conditioning block through the per-frame bias, previous activations in order -- and
the slot permutation must agree with the reference's PosEmbedding channel layout (embedding.py:96-108, pinned for the
oracle by tests/golden/ops.pt).  lab4d_mlp_describe is a host function of the library (no GPU needed)."""
import torch

from lab4d_amd import mlp, synthetic
from oracle import lab4d_oracle as O


def all_weights():
    P = synthetic.add_dense_weights(synthetic.make_weights(0))
    P.update({"bg." + k: v for k, v in synthetic.make_bg_weights(0).items()})
    P.update({"d6." + k: v for k, v in synthetic.make_weights(0, motion="dense").items() if k.startswith("warp.")})  # fg_motion "dense"
    return P


CASES = [(mlp.NET_FG_BASE, ""), (mlp.NET_FG_COLOR, ""), (mlp.NET_VIS, ""), (mlp.NET_FEAT, ""), (mlp.NET_SKIN, ""),
         (mlp.NET_DENSE, "warp.post_warp.forward_map."), (mlp.NET_DENSE, "warp.post_warp.backward_map."),
         (mlp.NET_BG_BASE, "bg."), (mlp.NET_BG_COLOR, "bg."), (mlp.NET_DENSE6, "d6.warp.forward_map."), (mlp.NET_DENSE6, "d6.warp.backward_map.")]


def test_every_reference_weight_column_is_consumed_exactly_once():
    P = all_weights()
    for net, prefix in CASES:
        d = mlp.describe(net)
        bds = mlp.bindings(net, prefix)
        assert len(bds) == d.n_layers, (net, len(bds), d.n_layers)
        for layer, bd in enumerate(bds):
            W = P[bd.wname]
            L = d.layers[layer]
            assert W.shape[0] == L.mout, (bd.wname, W.shape, L.mout)
            cm = mlp.col_map(net, layer, "cpu").tolist()
            assert len(cm) == L.ke + L.kin
            used = [c for c in cm if c >= 0]
            assert len(used) == len(set(used)), bd.wname
            cond = list(range(bd.cond[0], bd.cond[0] + bd.cond[1])) if bd.cond else []
            assert not set(used) & set(cond), bd.wname
            assert sorted(used + cond) == list(range(W.shape[1])), (bd.wname, W.shape[1], len(used), len(cond))
            assert bool(L.pf_bias) == bool(bd.cond), bd.wname


def test_affine_form_skin_nets_bind_the_reference_layers_two_and_final():
    """LAB4D_NET_SKIN_A / _SKIN18_A: layers 0 / 1 are the reference's linear_2 / linear_final with identity column maps (the 64 "embedding" slots
    are the hidden features of the folded linear_1, in order); the folded layer's columns are consumed by warping.skin_affine_table."""
    for n_bones, net_old in ((25, mlp.NET_SKIN), (18, mlp.NET_SKIN18)):
        net = mlp.skin_net_for(n_bones, affine=True)
        assert net in (mlp.NET_SKIN_A, mlp.NET_SKIN18_A) and mlp.skin_net_for(n_bones) == net_old
        d, old = mlp.describe(net), mlp.describe(net_old)
        assert d.emb_kind == 2 and d.n_layers == old.n_layers - 1 and d.c_out == n_bones and d.ke == old.layers[0].mout_pad == 64
        bd, bd_old = mlp.bindings(net), mlp.bindings(net_old)
        assert [b.wname for b in bd] == [b.wname for b in bd_old[1:]]
        for layer in range(d.n_layers):
            L = d.layers[layer]
            assert (L.mout, L.relu, L.pf_bias) == (old.layers[layer + 1].mout, old.layers[layer + 1].relu, 0)
            assert mlp.col_map(net, layer, "cpu").tolist() == list(range(L.ke + L.kin))


def test_slot_permutation_matches_the_reference_embedding_layout():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(9, 3, generator=g)
    for n_freq in (6, 8, 10, 12):
        ref = O.pos_embedding(x, n_freq)  # reference channel order
        ke = (6 * n_freq + 3 + 31) // 32 * 32
        chan = mlp.posenc_slot_to_ref_channel(n_freq, ke)
        assert sorted(c for c in chan if c >= 0) == list(range(6 * n_freq + 3))
        for slot, c in enumerate(chan):
            if slot < 6 * n_freq:  # the kernel's slot contents: pairs (sin, cos) of one (frequency, axis)
                pair, t = slot >> 1, slot & 1
                f, a = pair // 3, pair % 3
                val = torch.sin(x[:, a] * 2.0 ** f) if t == 0 else torch.cos(x[:, a] * 2.0 ** f)
            elif slot < 6 * n_freq + 3:
                val = x[:, slot - 6 * n_freq]
            else:
                assert c == -1
                continue
            assert torch.allclose(ref[:, c], val, atol=1e-6), (n_freq, slot, c)


def test_first_layer_through_the_column_map_equals_the_reference_layer():
    """W[:, col_map] applied to the slot-ordered embedding + the per-frame bias == F.linear on [posenc | code] (CondMLP, base.py:139-146)."""
    P = all_weights()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(5, 3, generator=g)
    for net, prefix, n_freq, wname, code_name in [(mlp.NET_FG_BASE, "", 10, "basefield.linear_1.0", "basefield.inst_embedding.mapping.weight"),
                                                  (mlp.NET_VIS, "", 10, "vis_mlp.basefield.linear_1.0", "vis_mlp.basefield.inst_embedding.mapping.weight"),
                                                  (mlp.NET_BG_BASE, "bg.", 6, "bg.basefield.linear_1.0", "bg.basefield.inst_embedding.mapping.weight")]:
        W, b, code = P[wname + ".weight"], P[wname + ".bias"], P[code_name][:1]
        ref = torch.nn.functional.linear(torch.cat([O.pos_embedding(x, n_freq), code.expand(5, -1)], -1), W, b)
        cm = mlp.col_map(net, 0, "cpu")
        emb_ref = O.pos_embedding(x, n_freq)
        chan = mlp.posenc_slot_to_ref_channel(n_freq, len(cm))
        slots = torch.stack([emb_ref[:, c] if c >= 0 else torch.zeros(5) for c in chan], -1)  # what the kernel computes per slot
        Wk = torch.stack([W[:, c] if c >= 0 else torch.zeros(W.shape[0]) for c in cm.tolist()], -1)  # what lab4d_mlp_pack gathers
        got = slots @ Wk.t() + mlp.pf_bias_of(net, 0, W, code) + b
        assert torch.allclose(got, ref, atol=1e-5), wname
