"""Host-side batch producers (vimacsa_dataset.MACSADataset, iaog_dataset.IAOGDataset) and the precomputed-feature cache:
the reference's tuple layouts (vimacsa_dataset.py:202, iaog_dataset.py:99-103), dtypes, prompts, label mapping and ROI
padding, checked with a recording tokenizer and an in-memory image loader (no torchvision, no image files)."""
import numpy as np
import pandas as pd
import pytest
import torch


class RecTokenizer:
    """records the prompts; ids = byte values, so that the encoded text can be read back"""
    pad_token_id = 1

    def __init__(self):
        self.calls = []

    def __call__(self, first, second=None, max_length=170, truncation=None, padding=None, return_token_type_ids=False,
                 return_tensors=None):
        self.calls.append((first, second))
        ids = [0] + [3 + (b % 200) for b in first.encode()][:max_length - 2] + [2]
        n = len(ids)
        ids = ids + [self.pad_token_id] * (max_length - n)
        return {"input_ids": ids, "token_type_ids": [0] * max_length, "attention_mask": [1] * n + [0] * (max_length - n)}


def _loader(path):
    if path.endswith("broken.png"):
        raise OSError("unreadable")
    seed = sum(path.encode()) % 997
    return torch.from_numpy(np.random.default_rng(seed).integers(0, 256, size=(3, 300, 400), dtype=np.uint8))


def _frames():
    data = pd.DataFrame({
        "comment": ["Phong_sach dep", "Do an ngon"],
        "list_img": [["a.png", "broken.png", "c.png"], []],
        "x": [0, 0],
        "text_img_label": [["Room#Positive", "Public_area#Negative", "Room#Neutral"], ["Food#Positive"]],
        "iaog_labels": [["sach#Room", "dep#Room", "rong#Public_area", "bad"], []],
    })
    roi_df = pd.DataFrame({"file_name": ["a.png", "a.png", "c.png"], "x1": [10, 0, 600], "x2": [200, 0, 700],
                           "y1": [20, 5, 100], "y2": [380, 5, 900], "label": ["bed", "x", "pool"]})
    return data, roi_df, {"a.png": ["bed_room", "Window"], "c.png": ["pool"]}, {"a.png": ["bed"]}


def test_macsa_dataset_tuple_layout():
    from vimacsa_dataset import MACSADataset
    data, roi_df, tags_i, tags_r = _frames()
    tok = RecTokenizer()
    ds = MACSADataset(data, tok, "/imgs", roi_df, tags_i, tags_r, num_img=2, num_roi=3, image_loader=_loader)
    assert len(ds) == 2
    t_img, roi_img, coors, ids, types, masks, added, labels, text = ds[0]
    assert t_img.shape == (2, 3, 224, 224) and t_img.dtype == torch.float32
    assert roi_img.shape == (2, 3, 3, 224, 224) and roi_img.dtype == torch.float64      # numpy default in the reference
    assert coors.shape == (2, 3, 4) and coors.dtype == torch.float64
    assert ids.shape == types.shape == masks.shape == (6, 170) and added.shape == (6, 219) and bool((added == 1).all())
    assert text == "Phong_sach dep"
    # labels in aspect order; first annotation of an aspect wins; missing aspects -> None (0)
    assert labels.tolist() == [0, 0, 3, 0, 0, 1]
    # prompts: "<aspect> </s></s> <text>" and " <image tags> </s></s>  <roi tags>", lower-cased, '_' -> ' '
    first, second = tok.calls[5]
    assert first == "public area </s></s> phong sach dep"
    # tags of the first num_img photos only; the reference's list(set(...)) order is arbitrary -- here: sorted
    assert second == " window , bed room </s></s>  bed"
    # photo 0: two boxes (the second is empty -> zero crop, but its box is kept), third slot zero-padded
    assert t_img[0].abs().sum() > 0 and roi_img[0, 0].abs().sum() > 0
    assert roi_img[0, 1].abs().sum() == 0 and roi_img[0, 2].abs().sum() == 0
    assert torch.allclose(coors[0, 0], torch.tensor([10, 200, 20, 380], dtype=torch.float64) / 512)
    assert torch.allclose(coors[0, 1], torch.tensor([0, 0, 5, 5], dtype=torch.float64) / 512) and coors[0, 2].abs().sum() == 0
    # photo 1 is unreadable: zero image, no boxes -> zero crops and boxes
    assert t_img[1].abs().sum() == 0 and roi_img[1].abs().sum() == 0 and coors[1].abs().sum() == 0
    # the normalisation is ImageNet mean/std of the resized uint8 photo
    ref = torch.nn.functional.interpolate(_loader("/imgs/a.png").unsqueeze(0).float(), size=(224, 224), mode="bilinear",
                                          antialias=True).squeeze(0).round().clamp(0, 255) / 255
    assert torch.allclose(t_img[0, 1], (ref[1] - 0.456) / 0.224, atol=1e-6)
    # review without photos: tags 'empty', zero tensors
    t2 = ds[1]
    assert t2[0].abs().sum() == 0 and tok.calls[-1][1] == " empty </s></s>  empty" and t2[7].tolist() == [0, 3, 0, 0, 0, 0]


def test_iaog_dataset_tuple_layout():
    from iaog_dataset import IAOGDataset
    data, roi_df, tags_i, tags_r = _frames()
    tok = RecTokenizer()
    ds = IAOGDataset(data, tok, "/imgs", roi_df, tags_i, tags_r, num_img=3, num_roi=2, max_len_decoder=12, image_loader=_loader)
    assert len(ds) == 2                      # review 0: Room {sach, dep}, Public_area {rong}; review 1: none
    items = {ds.samples[i]["target_aspect"]: ds[i] for i in range(2)}
    t_img, roi_img, coors, labels, dec_ids, enc_ids, enc_type, enc_mask, added, aspect, text = items["Room"]
    assert t_img.shape == (3, 3, 224, 224) and roi_img.shape == (3, 2, 3, 224, 224)
    assert roi_img.dtype == torch.float32 and coors.dtype == torch.float32
    assert labels.shape == dec_ids.shape == (12,) and enc_ids.shape == (170,) and added.shape == (219,)
    assert aspect == "Room" and text == "Phong_sach dep"
    dec_prompt = [c for c in tok.calls if c[1] is None][0][0]
    assert dec_prompt in ("room dep , sach", "public area rong")             # "<aspect> <sorted words>"
    # labels: decoder ids shifted left; last position and pads -> -100
    n = int((dec_ids != tok.pad_token_id).sum())
    assert labels[:n - 1].tolist() == dec_ids[1:n].tolist() and (labels[n - 1:] == -100).all()
    # iaog_dataset.py:139: x1 = max(0, x1), x2 = min(rows, x2) -- c.png's box x 600..700 on a 300-row photo gives the empty
    # crop [600:300]; the box is /512 and clipped to [0, 1]
    assert roi_img[2, 0].abs().sum() == 0
    assert coors[2, 0].tolist() == pytest.approx([1.0, 300 / 512, 100 / 512, 400 / 512])


def test_iaog_dataset_honours_max_seq_length_and_list_aspect():
    """--max_seq_length / --list_aspect of run_pretraining_fcmf.py (reference :60,82) reach the dataset"""
    from iaog_dataset import IAOGDataset
    data, roi_df, tags_i, tags_r = _frames()
    ds = IAOGDataset(data, RecTokenizer(), "/imgs", roi_df, tags_i, tags_r, num_img=1, num_roi=1, max_len_decoder=8,
                     image_loader=_loader, max_seq_length=96, list_aspect=["Room"])
    assert [s["target_aspect"] for s in ds.samples] == ["Room"]              # Public_area is not in the list
    item = ds[0]
    assert item[5].shape == (96,) and item[8].shape == (96 + 49,) and item[4].shape == (8,)


def test_feature_cache_roundtrip(tmp_path):
    from feature_cache import FeatureCache, FeatureCacheWriter
    from vimacsa_dataset import MACSADataset
    n, NI, NR = 3, 2, 4
    g = torch.Generator().manual_seed(0)
    vis, roi = torch.randn(n, NI, 49, 2048, generator=g), torch.randn(n, NI, NR, 2048, generator=g)
    coors = torch.rand(n, NI, NR, 4, generator=g, dtype=torch.float64)
    w = FeatureCacheWriter(str(tmp_path / "fc"), n, NI, NR)
    w.append(vis[:2], roi[:2], coors[:2]); w.append(vis[2:], roi[2:], coors[2:])
    w.close()
    fc = FeatureCache(str(tmp_path / "fc"))
    assert len(fc) == n
    v, r, c = fc[1]
    assert v.dtype == torch.bfloat16 and v.shape == (NI, 49, 2048) and r.shape == (NI, NR, 2048)
    assert torch.equal(v, vis[1].bfloat16()) and torch.equal(r, roi[1].bfloat16()) and torch.allclose(c, coors[1].float())
    # a dataset on top of the cache yields features in the pixel positions of the tuple
    data, roi_df, tags_i, tags_r = _frames()
    ds = MACSADataset(data, RecTokenizer(), "/imgs", roi_df, tags_i, tags_r, num_img=NI, num_roi=NR, feature_cache=fc)
    item = ds[1]
    assert torch.equal(item[0], vis[1].bfloat16()) and item[3].shape == (6, 170)


def test_default_image_loader_is_loud_without_torchvision():
    from review_batches import default_image_loader
    try:
        import torchvision  # noqa: F401
        pytest.skip("torchvision present")
    except ImportError:
        with pytest.raises(RuntimeError, match="torchvision"):
            default_image_loader("/nonexistent.png")
