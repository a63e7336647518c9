"""MACSADataset: the fine-tuning batch producer, drop-in for the reference's vimacsa_dataset.py (same constructor
arguments, same 9-tuple per review: vimacsa_dataset.py:14-202).

    (t_img_features [NI,3,224,224] f32, roi_img_features [NI,NR,3,224,224] f64, roi_coors [NI,NR,4] f64,
     input_ids [6,170], token_type_ids [6,170], attention_mask [6,170], added_input_mask [6,219], label_ids [6], text)

One row per aspect category in the fixed order Location, Food, Room, Facilities, Service, Public_area; labels
None/Negative/Neutral/Positive -> 0..3, aspects without an annotation -> None (:67-82,112).  With `feature_cache=`
(feature_cache.FeatureCache) the first two entries are the precomputed ResNet-152 features [NI,49,2048] / [NI,NR,2048]
instead of pixels (BASELINE.json's configs; run the driver with --precomputed_features).
"""
import torch

from review_batches import ASPECTS, POLARITY, ReviewProducer, display_name


class MACSADataset(torch.utils.data.Dataset):
    def __init__(self, data, tokenizer, img_folder, roi_df, dict_image_aspect, dict_roi_aspect, num_img, num_roi,
                 image_loader=None, feature_cache=None):
        self.data = data
        self.ASPECT = list(ASPECTS)
        self.pola_to_num = dict(POLARITY)
        self.num_img, self.num_roi = num_img, num_roi
        self.producer = ReviewProducer(tokenizer, img_folder, roi_df, dict_image_aspect, dict_roi_aspect, num_img, num_roi,
                                       image_loader=image_loader, feature_cache=feature_cache, roi_dtype=torch.float64)

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, idx):
        row = self.data.iloc[idx, :].values
        text, photos, annotations = row[0], row[1], row[3]
        polarity = {}
        for item in annotations:                          # "Aspect#Polarity"; the first mention of an aspect wins
            asp, pol = item.split("#")
            polarity.setdefault(display_name(asp), pol)
        tags = self.producer.visual_tags(photos)
        ids, types, masks, added, labels = [], [], [], [], []
        for asp in self.ASPECT:
            i, t, m, a = self.producer.encode(asp, text, tags)
            ids.append(i); types.append(t); masks.append(m); added.append(a)
            labels.append(self.pola_to_num[polarity.get(display_name(asp), "None")])
        vis, roi, coors = self.producer.visual(idx, photos)
        return (vis, roi, coors, torch.stack(ids), torch.stack(types), torch.stack(masks), torch.stack(added),
                torch.tensor(labels), text)
