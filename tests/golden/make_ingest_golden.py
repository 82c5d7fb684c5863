"""Golden vectors for the batch-ingestion row from the REFERENCE's own dataloader: the real
`lab4d.dataloader.vidloader.VidDataset` is run on a synthetic video written to a temp dir in the reference's on-disk layout
(the .npy names `construct_data_list` derives, vidloader.py:70-121).  Only outputs, pixel coordinates and the seed of the synthetic
video are stored (oracle/ingest_oracle.synthetic_video regenerates the inputs).  Build container only:
    python tests/golden/make_ingest_golden.py        -> tests/golden/ingest.pt"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
OUT_DIR = os.environ.get("LAB4D_GOLDEN_OUT", HERE)
from oracle import ingest_oracle as IO, ref_shim  # noqa: E402


def write_dataset(root, video, seq="seq-0000", prefix="crop-24", feature_type="cse"):
    T = video["rgb"].shape[0]
    d = lambda kind: os.path.join(root, kind, "Full-Resolution", seq)  # noqa: E731
    for kind in ["JPEGImages", "JPEGImagesRaw", "Annotations", "Depth", "Features", "Cameras"] + ["FlowFW_%d" % k for k in video["flowfw"]] + \
            ["FlowBW_%d" % k for k in video["flowbw"]]:
        os.makedirs(d(kind), exist_ok=True)
    reflist = []
    for t in range(T):
        p = os.path.join(d("JPEGImages"), "%05d.jpg" % t)
        open(p, "w").close()
        open(os.path.join(d("JPEGImagesRaw"), "%05d.jpg" % t), "w").close()
        reflist.append(p)
    np.save(os.path.join(d("JPEGImages"), prefix + ".npy"), video["rgb"])
    np.save(os.path.join(d("Annotations"), prefix + ".npy"), video["mask"])
    np.save(os.path.join(d("Annotations"), prefix + "-crop2raw.npy"), video["crop2raw"])
    np.save(os.path.join(d("Annotations"), prefix + "-is_detected.npy"), video["is_detected"])
    np.save(os.path.join(d("Depth"), prefix + ".npy"), video["depth"])
    np.save(os.path.join(d("Features"), "%s-%s-01.npy" % (prefix, feature_type)), video["feature"])
    for k, v in video["flowfw"].items():
        np.save(os.path.join(d("FlowFW_%d" % k), prefix + ".npy"), v)
    for k, v in video["flowbw"].items():
        np.save(os.path.join(d("FlowBW_%d" % k), prefix + ".npy"), v)
    return reflist, prefix, feature_type


def _t(v):
    a = np.asarray(v)
    return torch.from_numpy(a.copy() if a.ndim else a.reshape(1).copy()).reshape(a.shape)


def main():
    ref_shim.load()
    import importlib
    vl = importlib.import_module("lab4d.dataloader.vidloader")
    seed, T, H, W, N = 5, 5, 24, 24, 16
    video = IO.synthetic_video(seed, T=T, H=H, W=W)
    out = {"meta": {"seed": seed, "T": T, "H": H, "W": W, "N": N, "deltas": [1, 2]}, "read_raw": [], "load_data": []}
    with tempfile.TemporaryDirectory() as root:
        reflist, prefix, ft = write_dataset(root, video)
        opts = {"delta_list": [2], "data_prefix": prefix, "feature_type": ft, "pixels_per_image": N, "load_pair": True}
        ds = vl.VidDataset(opts, reflist, dataid=3, ks=[24.0, 24.0, 12.0, 12.0], raw_size=[24, 24])
        assert tuple(ds.img_size) == (H, W)
        r = np.random.default_rng(11)
        # read_raw on chosen (frame, delta) incl. backward flow, the last frame, border pixels (feature clip) and repeated pixels
        cases = [(0, 1), (1, -1), (2, 2), (4, -2), (3, 1), (4, -1), (0, 2)]
        for im0, delta in cases:
            xy = np.stack([r.integers(0, W, N), r.integers(0, H, N)], -1)
            xy[0] = (0, 0); xy[1] = (W - 1, H - 1); xy[2] = (W - 1, 0); xy[3] = xy[4]
            d = ds.read_raw(im0, delta, rand_xy=xy)
            out["read_raw"].append({"im0idx": im0, "delta": delta, "xy": torch.from_numpy(xy.copy()),
                                    "out": {k: _t(v) for k, v in d.items()}})
        # load_data: numpy's global RNG drives sample_delta / the RangeSampler permutation; the drawn delta and pixels are recovered from
        # the output itself (hxy; frameid_sub) so the fixture pins the pair stacking and the index arithmetic of sample_xy
        for idx in [0, 2, 3]:
            np.random.seed(100 + idx)
            ds.idx_sampler.init_queue()
            q = ds.idx_sampler.sample_queue.copy()
            d = ds.load_data(idx)
            out["load_data"].append({"im0idx": idx, "queue_head": torch.from_numpy(q[: 2 * N].copy()),
                                     "out": {k: _t(v) for k, v in d.items()}})
    torch.save(out, os.path.join(OUT_DIR, "ingest.pt"))
    print("wrote", os.path.join(OUT_DIR, "ingest.pt"), os.path.getsize(os.path.join(OUT_DIR, "ingest.pt")), "bytes")


if __name__ == "__main__":
    main()
