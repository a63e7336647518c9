"""Host-side batch producer shared by `vimacsa_dataset.MACSADataset` and `iaog_dataset.IAOGDataset`: turns one hotel
review (text + up to num_img photos + up to num_roi detected regions per photo) into the tensors the FCMF step consumes,
in the reference's tuple layout (vimacsa_dataset.py:202, iaog_dataset.py:103).

What it restates from the reference (behaviour, not code):
  * the two-segment encoder prompt "<aspect> </s></s> <review>" + " <image tags> </s></s>  <roi tags>", lower-cased,
    '_' -> ' ', tokenised to 170 positions with truncation of the first segment (vimacsa_dataset.py:97-104);
  * tags = union of the per-photo ResNet labels of the first num_img photos, 'empty' when none (:47-65);
  * photos: RGB -> 224x224 (antialiased bilinear) -> float -> ImageNet mean/std (:26-31); an unreadable photo is a zero
    image (:138-142); ROIs are crops one_image[:, x1:x2, y1:y2] of the ORIGINAL photo through the same transform, boxes
    (x1,x2,y1,y2)/512 clipped to [0,1] (:158-172), zero crops / zero boxes pad up to num_roi (:174-177);
  * dtypes as the reference leaves them: MACSA ROI crops and boxes float64 (numpy default), IAOG float32.
MI355X-first addition: a `FeatureCache` (feature_cache.py) replaces the pixel tensors by precomputed ResNet-152 features
(BASELINE.json's configs) -- same tuple positions, so the drivers do not care which one they got.

Image decoding is pluggable (`image_loader(path) -> uint8 tensor [3, H, W]`): the default uses torchvision.io when it is
importable and raises a clear error otherwise (torchvision is not part of this environment); there is no silent stand-in.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

ASPECTS = ['Location', 'Food', 'Room', 'Facilities', 'Service', 'Public_area']
POLARITY = {"None": 0, "Negative": 1, "Neutral": 2, "Positive": 3}
IMAGENET_MEAN = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
IMAGENET_STD = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
SEQ_LEN, PATCHES = 170, 49


def default_image_loader(path):
    try:
        from torchvision.io import read_image, ImageReadMode
    except ImportError as e:
        raise RuntimeError("decoding image files needs torchvision.io (not installed here): pass image_loader=... "
                           "or use a FeatureCache of precomputed ResNet-152 features") from e
    return read_image(path, mode=ImageReadMode.RGB)


def to_crop(img_u8, size=224):
    """uint8 [3,H,W] -> normalised float32 [3,size,size] (Resize(antialias) + ConvertImageDtype + Normalize)"""
    x = F.interpolate(img_u8.unsqueeze(0).float(), size=(size, size), mode="bilinear", antialias=True, align_corners=False)
    x = x.squeeze(0).round().clamp_(0, 255) / 255.0 if img_u8.dtype == torch.uint8 else x.squeeze(0)
    return (x - IMAGENET_MEAN) / IMAGENET_STD


def display_name(aspect):
    return "Public area" if "_" in aspect else aspect


class ReviewProducer:
    def __init__(self, tokenizer, img_folder, roi_df, dict_image_aspect, dict_roi_aspect, num_img, num_roi,
                 image_loader=None, feature_cache=None, roi_dtype=torch.float64, clamp_boxes=False, crop_size=224, seq_len=None):
        self.tokenizer, self.img_folder, self.roi_df = tokenizer, img_folder, roi_df
        self.tags_img, self.tags_roi = dict_image_aspect, dict_roi_aspect
        self.num_img, self.num_roi = num_img, num_roi
        self.load = image_loader or default_image_loader
        self.cache = feature_cache
        self.roi_dtype, self.clamp_boxes, self.size = roi_dtype, clamp_boxes, crop_size
        self.seq_len = int(seq_len) if seq_len else SEQ_LEN

    # ---- text side -------------------------------------------------------------------------------------------
    def visual_tags(self, photos):
        img, roi = [], []
        for name in list(photos or [])[:self.num_img]:
            img.extend(self.tags_img.get(name, []) or [])
            roi.extend(self.tags_roi.get(name, []) or [])
        return (sorted(set(img)) or ['empty']), (sorted(set(roi)) or ['empty'])

    def encode(self, aspect, text, tags):
        """-> (input_ids, token_type_ids, attention_mask, added_mask) of the two-segment prompt"""
        first = f"{display_name(aspect)} </s></s> {text}".lower().replace('_', ' ')
        second = f" {' , '.join(tags[0])} </s></s>  {' , '.join(tags[1])}".lower().replace('_', ' ')
        tok = self.tokenizer(first, second, max_length=self.seq_len, truncation='only_first', padding='max_length',
                             return_token_type_ids=True)
        as_t = lambda k: torch.as_tensor(tok[k]).reshape(-1)
        return as_t('input_ids'), as_t('token_type_ids'), as_t('attention_mask'), torch.ones(self.seq_len + PATCHES, dtype=torch.long)

    # ---- image side ------------------------------------------------------------------------------------------
    def _boxes_of(self, name):
        if self.roi_df is None:
            return []
        rows = self.roi_df[self.roi_df['file_name'] == name][:self.num_roi]
        return [tuple(int(v) for v in rows.iloc[i, 1:5].values) for i in range(rows.shape[0])]

    def pixels(self, photos):
        """-> t_img [num_img,3,S,S] float32, roi_img [num_img,num_roi,3,S,S], roi_coors [num_img,num_roi,4]"""
        S = self.size
        t_img = torch.zeros(self.num_img, 3, S, S)
        roi_img = torch.zeros(self.num_img, self.num_roi, 3, S, S, dtype=self.roi_dtype)
        coors = torch.zeros(self.num_img, self.num_roi, 4, dtype=self.roi_dtype)
        for i, name in enumerate(list(photos or [])[:self.num_img]):
            try:
                photo = self.load(os.path.join(self.img_folder, name))
                t_img[i] = to_crop(photo, S)
            except (OSError, RuntimeError, ValueError):
                photo = torch.zeros(3, S, S, dtype=torch.uint8)            # unreadable photo: zero image, zero crops
            for r, (x1, x2, y1, y2) in enumerate(self._boxes_of(name)):
                if self.clamp_boxes:
                    x1, x2 = max(0, x1), min(photo.shape[1], x2)
                    y1, y2 = max(0, y1), min(photo.shape[2], y2)
                crop = photo[:, x1:x2, y1:y2]
                if crop.numel() > 0:
                    roi_img[i, r] = to_crop(crop, S).to(self.roi_dtype)
                coors[i, r] = torch.tensor(np.clip(np.array([x1, x2, y1, y2]) / 512.0, 0.0, 1.0), dtype=self.roi_dtype)
        return t_img, roi_img, coors

    def visual(self, index, photos):
        """pixel tensors, or the cached features of review `index` in the same tuple positions"""
        if self.cache is not None:
            return self.cache[index]
        return self.pixels(photos)
