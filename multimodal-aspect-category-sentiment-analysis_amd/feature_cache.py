"""Precomputed ResNet-152 feature cache (BASELINE.json's configs train on precomputed features).

On disk (one directory):
    meta.json              {"n": N, "num_img": NI, "patches": 49, "num_roi": NR, "dim": 2048, "dtype": "bfloat16"}
    vis.bf16   [N, NI, 49, 2048]   image-patch features      (raw little-endian bfloat16, row-major)
    roi.bf16   [N, NI, NR, 2048]   region features
    coors.f32  [N, NI, NR, 4]      region boxes
Review i is three contiguous byte ranges (NI*49*4 KiB, NI*NR*4 KiB, ...): the loader memory-maps the files, so a batch
is a handful of large sequential reads and the tensors arrive in the dtype the MFMA path consumes (no float64 / float32
detour through the host).  `build` runs the HIP trunk (resnet_utils.extract_features, BatchNorm in eval mode) over a
pixel-producing dataset and appends the features batch by batch.
"""
import json
import os

import numpy as np
import torch

META = "meta.json"


class FeatureCache:
    def __init__(self, path):
        with open(os.path.join(path, META)) as f:
            self.meta = json.load(f)
        m = self.meta
        self.n = m["n"]
        shp = lambda *s: (m["n"],) + s
        self.vis = np.memmap(os.path.join(path, "vis.bf16"), dtype=np.uint16, mode="r", shape=shp(m["num_img"], m["patches"], m["dim"]))
        self.roi = np.memmap(os.path.join(path, "roi.bf16"), dtype=np.uint16, mode="r", shape=shp(m["num_img"], m["num_roi"], m["dim"]))
        self.coors = np.memmap(os.path.join(path, "coors.f32"), dtype=np.float32, mode="r", shape=shp(m["num_img"], m["num_roi"], 4))

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        bf = lambda a: torch.from_numpy(np.array(a[i])).view(torch.bfloat16)      # one contiguous read per array
        return bf(self.vis), bf(self.roi), torch.from_numpy(np.array(self.coors[i]))


class FeatureCacheWriter:
    def __init__(self, path, n, num_img, num_roi, patches=49, dim=2048):
        os.makedirs(path, exist_ok=True)
        self.path, self.meta = path, dict(n=n, num_img=num_img, patches=patches, num_roi=num_roi, dim=dim, dtype="bfloat16")
        self.files = {k: open(os.path.join(path, k), "wb") for k in ("vis.bf16", "roi.bf16", "coors.f32")}
        self.written = 0

    def append(self, vis, roi, coors):
        """vis [b,NI,49,2048], roi [b,NI,NR,2048] (any float dtype, any device), coors [b,NI,NR,4]"""
        to_u16 = lambda t: t.detach().to(torch.bfloat16).contiguous().cpu().view(torch.uint16).numpy()
        self.files["vis.bf16"].write(to_u16(vis).tobytes())
        self.files["roi.bf16"].write(to_u16(roi).tobytes())
        self.files["coors.f32"].write(coors.detach().float().contiguous().cpu().numpy().tobytes())
        self.written += vis.shape[0]

    def close(self):
        for f in self.files.values():
            f.close()
        assert self.written == self.meta["n"], (self.written, self.meta["n"])
        with open(os.path.join(self.path, META), "w") as f:
            json.dump(self.meta, f)


def build(dataset, resnet_img, resnet_roi, path, batch_size=8, device="cuda"):
    """features of every review of a pixel-producing dataset (MACSADataset / IAOGDataset tuples: pixels first)"""
    from fcmf_framework.resnet_utils import extract_features
    first = dataset[0]
    w = FeatureCacheWriter(path, len(dataset), first[0].shape[0], first[1].shape[1])
    resnet_img.eval(); resnet_roi.eval()
    for s in range(0, len(dataset), batch_size):
        items = [dataset[i] for i in range(s, min(len(dataset), s + batch_size))]
        t_img = torch.stack([it[0] for it in items]).to(device)
        roi_img = torch.stack([it[1] for it in items]).to(device)
        vis, roi = extract_features(resnet_img, resnet_roi, t_img, roi_img)
        w.append(vis, roi, torch.stack([it[2] for it in items]))
    w.close()
    return FeatureCache(path)
