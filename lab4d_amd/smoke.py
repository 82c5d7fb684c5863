"""One tiny invocation of the hot path on cuda:0, checked against the CPU oracle."""
import torch


def run():
    from oracle import lab4d_oracle as O
    from . import render_utils as RU
    g = torch.Generator().manual_seed(0)
    M, N, D = 2, 8, 16
    fd = {"density": torch.rand(M, N, D, 1, generator=g) * 30, "rgb": torch.rand(M, N, D, 3, generator=g),
          "vis": torch.randn(M, N, D, 1, generator=g)}
    fd["density_fg"] = fd["density"]
    deltas = torch.rand(M, N, D, 1, generator=g) * 0.03
    ref = O.render_pixel(fd, deltas)
    dev = RU.render_pixel({k: v.cuda() for k, v in fd.items()}, deltas.cuda())
    for k in ref:
        assert torch.allclose(dev[k].cpu(), ref[k], rtol=1e-4, atol=1e-5), k
