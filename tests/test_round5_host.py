"""CPU tests of round-5 host logic: the batched random draws of the whole-step graph, and the checker the eval-path index parity rests on
(fixture_utils.check_index_mismatches must REJECT an index difference that is not a verified one-bin tie)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_batched_draws_have_the_per_chunk_draws_distribution():
    """bench.draw_rng_all: per chunk a uniformly random subset of the rays in random order (no repeats) and 1,024 DISTINCT uniform sample indices --
    what draw_rng gives one chunk at a time (randperm prefix / distinct_indices)."""
    import bench
    g = torch.Generator().manual_seed(3)
    C, M, N, S = 4, 2, 4096, 2 * 4096 * 128
    outs = [{"eik_inds": torch.empty(M * N // 16, dtype=torch.int64), "match_perm": torch.empty(1024, dtype=torch.int64)} for _ in range(C)]
    bench.draw_rng_all(C, M, N, S, "cpu", g, outs)
    for o in outs:
        e, p = o["eik_inds"], o["match_perm"]
        assert e.numel() == M * N // 16 and int(e.min()) >= 0 and int(e.max()) < M * N and e.unique().numel() == e.numel()
        assert p.numel() == 1024 and int(p.min()) >= 0 and int(p.max()) < S and p.unique().numel() == 1024
    assert not torch.equal(outs[0]["eik_inds"], outs[1]["eik_inds"]) and not torch.equal(outs[0]["match_perm"], outs[1]["match_perm"])
    # the stacked form the captured step uses: rows of two (C, k) buffers
    stack = (torch.empty(C, M * N // 16, dtype=torch.int64), torch.empty(C, 1024, dtype=torch.int64))
    bench.draw_rng_all(C, M, N, S, "cpu", g, stack)
    assert all(stack[0][c].unique().numel() == M * N // 16 and stack[1][c].unique().numel() == 1024 for c in range(C))
    # small S: the permutation-prefix branch
    small = [{"eik_inds": torch.empty(1, dtype=torch.int64), "match_perm": torch.empty(96, dtype=torch.int64)} for _ in range(2)]
    bench.draw_rng_all(2, 2, 6, 96, "cpu", g, small)
    assert all(sorted(o["match_perm"].tolist()) == list(range(96)) for o in small)


def _tie_fixture():
    """A two-ray "fixture": 4 query points u, reference indices, and ONE near tie listed (ray 0 of frame 0, cdf entry 2 = 0.50000003 next to u[2] = 0.5)."""
    u = torch.tensor([0.0, 0.25, 0.5, 1.0])
    ref = torch.tensor([[[1, 2, 2, 4], [1, 1, 3, 4]]])  # (M=1, N=2, 4)
    g = {"meta": {"tie_window": 1e-4}, "u": u,
         "ties": {"band": torch.tensor([0]), "m": torch.tensor([0]), "n": torch.tensor([0]), "k": torch.tensor([2]), "cdf": torch.tensor([0.50000003])}}
    cdf_dev = torch.tensor([[[0.0, 0.2, 0.49999997, 0.9, 1.0], [0.0, 0.3, 0.45, 0.8, 1.0]]])
    return g, ref, cdf_dev


def test_index_checker_accepts_a_verified_tie_and_rejects_everything_else():
    from fixture_utils import check_index_mismatches
    g, ref, cdf_dev = _tie_fixture()
    # identical indices: nothing to verify
    assert check_index_mismatches(g, 0, ref.clone(), cdf_dev, tol=2e-6, ref=ref) == (0, 0.0)
    # ray 0, query 2 (u = 0.5): the device counts entry 2 (0.49999997 <= 0.5), the reference does not (0.50000003 > 0.5): a one-bin shift at the listed tie
    dev = ref.clone()
    dev[0, 0, 2] = 3
    n, gap = check_index_mismatches(g, 0, dev, cdf_dev, tol=2e-6, ref=ref)
    assert n == 1 and 0 < gap < 2e-6
    # the same shift with cdfs that differ by more than the floor: rejected
    far = cdf_dev.clone()
    far[0, 0, 2] = 0.4999
    with pytest.raises(AssertionError, match="fp32 floor|straddle"):
        check_index_mismatches(g, 0, dev, far, tol=2e-6, ref=ref)
    # a shift at an entry that is NOT a listed near tie (ray 1): rejected
    dev2 = ref.clone()
    dev2[0, 1, 1] = 2
    with pytest.raises(AssertionError, match="not within"):
        check_index_mismatches(g, 0, dev2, cdf_dev, tol=2e-6, ref=ref)
    # a two-bin difference: rejected
    dev3 = ref.clone()
    dev3[0, 0, 2] = 4
    with pytest.raises(AssertionError, match="more than one bin"):
        check_index_mismatches(g, 0, dev3, cdf_dev, tol=2e-6, ref=ref)


def test_cdf_of_weights_is_sample_pdf_s_cdf():
    """fixture_utils.cdf_of_weights = the cdf render_utils.sample_pdf forms (render_utils.py:203-207), as restated by the oracle."""
    from fixture_utils import cdf_of_weights
    from oracle import lab4d_oracle as O
    g = torch.Generator().manual_seed(1)
    w = torch.rand(7, 16, generator=g)
    bins = torch.sort(torch.rand(7, 15, generator=g), -1)[0]
    cdf = cdf_of_weights(w)
    # the oracle's searchsorted on ITS cdf must agree with searchsorted on this one for every query point
    u = torch.linspace(0, 1, 16)[None].expand(7, -1).contiguous()
    _, inds = O.sample_pdf(bins, w[:, 1:-1], 16, return_inds=True)
    assert torch.equal(torch.searchsorted(cdf, u, right=True), inds)
