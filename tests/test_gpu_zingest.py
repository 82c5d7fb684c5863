"""Batch ingestion on the device through the C ABI (lab4d_ingest_gather): bit-exact against the fixture the REAL
`VidDataset` generated (tests/golden/ingest.pt) and against the numpy oracle at a training-batch size."""
import os

import numpy as np
import pytest
import torch

from oracle import ingest_oracle as IO

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


def cache_of(video, dataid=0):
    from lab4d_amd import ingest
    return ingest.FrameCache(video["rgb"], video["mask"], video["depth"], video["flowfw"], video["flowbw"], video["feature"], video["crop2raw"],
                             video["is_detected"], dataid=dataid, device=DEV)


def same(dev_t, ref):
    a = dev_t.cpu().numpy()
    b = ref.numpy() if torch.is_tensor(ref) else np.asarray(ref)
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    return np.array_equal(a.view(np.uint8) if a.dtype.kind == "f" else a, b.view(np.uint8) if b.dtype.kind == "f" else b)


def test_gather_matches_the_reference_dataloader_bit_for_bit():
    from lab4d_amd import ingest
    g = torch.load(os.path.join(ROOT, "tests", "golden", "ingest.pt"), weights_only=False)
    m = g["meta"]
    video = IO.synthetic_video(m["seed"], T=m["T"], H=m["H"], W=m["W"])
    cache = cache_of(video, dataid=3)
    loader = ingest.DeviceVidLoader(cache, delta_list=[2], pixels_per_image=m["N"])
    # every read_raw case of the fixture in ONE launch
    items = [(0, c["im0idx"], c["delta"]) for c in g["read_raw"]]
    out = ingest.gather([cache], items, torch.stack([c["xy"] for c in g["read_raw"]]).to(DEV))
    for j, c in enumerate(g["read_raw"]):
        for k, ref in c["out"].items():
            assert same(out[k][j], ref), (k, c["im0idx"], c["delta"])
        one = loader.read_raw(c["im0idx"], c["delta"], c["xy"].to(DEV))
        assert all(same(one[k], ref) for k, ref in c["out"].items())
    # load_data: pair stacking, with the reference's drawn delta and pixels injected
    for c in g["load_data"]:
        ref, q = c["out"], c["queue_head"].numpy()
        xy0, xy1 = IO.sample_xy_from_idx(q[: m["N"]], m["H"]), IO.sample_xy_from_idx(q[m["N"]: 2 * m["N"]], m["H"])
        delta = int(ref["frameid_sub"][1] - ref["frameid_sub"][0])
        pair = loader.load_data(c["im0idx"], delta=delta, xy=[torch.from_numpy(xy0).to(DEV), torch.from_numpy(xy1).to(DEV)])
        for k, r in ref.items():
            assert same(pair[k], r), k


@pytest.mark.parametrize("dt", [np.float16, np.float32])
def test_training_batch_size_against_the_oracle_and_sampler_properties(dt):
    """256 frames x 16 pixels (the reference's default batch: imgs_per_gpu 128 pairs, pixels_per_image 16) out of a 256x256 video."""
    from lab4d_amd import ingest
    T, H, N = 9, 256, 16
    video = IO.synthetic_video(21, T=T, H=H, W=H, deltas=(1, 2, 4))
    if dt == np.float32:
        for k in ("rgb", "depth", "feature"):
            video[k] = video[k].astype(np.float32)
        video["flowfw"] = {d: v.astype(np.float32) for d, v in video["flowfw"].items()}
        video["flowbw"] = {d: v.astype(np.float32) for d, v in video["flowbw"].items()}
    cache = cache_of(video)
    gen = torch.Generator(device=DEV).manual_seed(5)
    loader = ingest.DeviceVidLoader(cache, delta_list=[2, 4], pixels_per_image=N, generator=gen)
    np.random.seed(0)
    idx = np.random.randint(0, len(loader), 128)
    batch = loader.load_batch(list(idx))
    assert batch["rgb"].shape == (128, 2, N, 3) and batch["feature"].shape == (128, 2, N, 16) and batch["hxy"].dtype == torch.float32
    fid = batch["frameid_sub"].cpu().numpy()
    xy = batch["hxy"][..., :2].cpu().numpy().astype(np.int64)
    for j in range(0, 128, 9):
        delta = int(fid[j, 1] - fid[j, 0])
        assert fid[j, 0] == idx[j] and delta in (1, 2, 4) and idx[j] % delta == 0
        ref = IO.load_pair(video, int(idx[j]), delta, xy[j, 0], xy[j, 1])
        for k in ("rgb", "mask", "vis2d", "depth", "flow", "flow_uct", "feature", "hxy", "crop2raw", "is_detected"):
            assert same(batch[k][j], ref[k]), (k, j)
    # RangeSampler: without replacement until the permutation is used up (vidloader.py:13-43)
    s = ingest.RangeSampler(1000, device=DEV, generator=gen)
    drawn = torch.cat([s.sample(100) for _ in range(10)])
    assert drawn.unique().numel() == 1000
    s.sample(100)
    assert s.curr_idx == 100  # re-permuted


def test_bad_arguments_fail_loudly():
    from lab4d_amd import ingest
    video = IO.synthetic_video(1, T=3, H=8, W=8, deltas=(1,))
    cache = cache_of(video)
    with pytest.raises(IndexError):
        ingest.gather([cache], [(0, 0, 1)], torch.tensor([[[8, 0]]], device=DEV))
    with pytest.raises(IndexError):
        ingest.gather([cache], [(0, 0, -1)], torch.tensor([[[0, 0]]], device=DEV))  # no backward flow into frame 0
    with pytest.raises(RuntimeError):
        ingest.gather([cache], [(0, 0, 1)], torch.zeros(1, 4, 3, device=DEV))


def test_vid_dataset_adapter_serves_load_data_from_the_frame_cache():
    """patch.vid_load_data (round 4, opt-in binding of VidDataset.load_data): a stand-in dataset object carrying the attributes the real
    VidDataset has after __init__ (mmap_list, crop2raw, is_detected, dataid, frame_info.frame_map, delta_list, pixels_per_image, load_pair,
    sample_delta) -> the pair comes back from the HBM-resident cache with the reference's keys / shapes / dtypes, and every value equals
    the numpy oracle's for the pixels the device sampler drew."""
    import types
    from lab4d_amd import patch
    T, H, N = 7, 32, 16
    video = IO.synthetic_video(33, T=T, H=H, W=H, deltas=(1, 2))
    ds = types.SimpleNamespace(mmap_list={k: video[k] for k in ("rgb", "mask", "depth", "flowfw", "flowbw", "feature")}, crop2raw=video["crop2raw"],
                               is_detected=video["is_detected"], dataid=5, frame_info=types.SimpleNamespace(frame_map=list(range(100, 100 + T))),
                               delta_list=[2], pixels_per_image=N, load_pair=True, sample_delta=lambda i: 2 if i % 2 == 0 and i + 2 < T else 1)
    for im0 in (0, 3, 4):
        out = patch.vid_load_data(ds, im0)
        delta = ds.sample_delta(im0)
        assert out["rgb"].shape == (2, N, 3) and out["hxy"].shape == (2, N, 3) and out["mask"].dtype == torch.bool and out["rgb"].is_cuda
        assert out["frameid_sub"].tolist() == [100 + im0, 100 + im0 + delta] and out["dataid"].tolist() == [5, 5]
        xy = out["hxy"][..., :2].cpu().numpy().astype(np.int64)
        ref = IO.load_pair(video, im0, delta, xy[0], xy[1])
        for k in ("rgb", "mask", "vis2d", "depth", "flow", "flow_uct", "feature", "hxy", "crop2raw", "is_detected"):
            assert same(out[k], ref[k]), (k, im0)
    assert ds._lab4d_device_loader is patch.device_loader_of(ds)  # uploaded once
