"""IAOGDataset: the IAOG pre-training batch producer, drop-in for the reference's iaog_dataset.py (same constructor,
same 11-tuple per sample: iaog_dataset.py:9-103).

One sample per (review, aspect) pair that has implicit-aspect opinion words ("word#Aspect" strings in the review's
`iaog_labels`): the decoder target is "<aspect> <sorted opinion words joined by ' , '>" (:37-57,88-96).

    (t_img_features [NI,3,224,224] f32, roi_img_features [NI,NR,3,224,224] f32, roi_coors [NI,NR,4] f32,
     labels [Ld] (decoder ids shifted left, last position and pads -> -100), dec_input_ids [Ld],
     enc_ids [170], enc_type [170], enc_mask [170], added_mask [219], target_aspect, text)
"""
import torch

from review_batches import ASPECTS, ReviewProducer


class IAOGDataset(torch.utils.data.Dataset):
    def __init__(self, data, tokenizer, img_folder, roi_df, dict_image_aspect, dict_roi_aspect, num_img=7, num_roi=4,
                 max_len_decoder=20, image_loader=None, feature_cache=None, max_seq_length=None, list_aspect=None):
        """max_seq_length / list_aspect: the driver's --max_seq_length / --list_aspect (the reference accepts both flags and
        hard-codes 170 positions and the six categories in its dataset: those are the defaults here)"""
        self.data, self.tokenizer = data, tokenizer
        self.num_img, self.num_roi, self.max_len_decoder = num_img, num_roi, max_len_decoder
        self.ASPECT = list(list_aspect) if list_aspect else list(ASPECTS)
        self.aspect2id = {a: i for i, a in enumerate(self.ASPECT)}
        self.producer = ReviewProducer(tokenizer, img_folder, roi_df, dict_image_aspect, dict_roi_aspect, num_img, num_roi,
                                       image_loader=image_loader, feature_cache=feature_cache, roi_dtype=torch.float32,
                                       clamp_boxes=True, seq_len=max_seq_length)
        self.samples = []
        for idx, row in self.data.iterrows():
            words_of = {}
            raw = row.get('iaog_labels', [])
            for item in (raw if isinstance(raw, list) else []):
                if '#' not in item:
                    continue
                word, aspect = (s.strip() for s in item.split('#')[:2])
                if aspect in self.aspect2id and word not in words_of.setdefault(aspect, []):
                    words_of[aspect].append(word)
            for aspect, words in words_of.items():
                self.samples.append({'original_idx': idx, 'target_aspect': aspect,
                                     'target_sentiment': ' , '.join(sorted(words))})

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        s = self.samples[idx]
        row = self.data.iloc[s['original_idx']]
        text, photos = row['comment'], row['list_img']
        tags = self.producer.visual_tags(photos)
        enc_ids, enc_type, enc_mask, added = self.producer.encode(s['target_aspect'], text, tags)
        dec_text = f"{s['target_aspect']} {s['target_sentiment']}".lower().replace('_', ' ')
        dec = self.tokenizer(dec_text, max_length=self.max_len_decoder, padding='max_length', truncation=True)
        dec_ids = torch.as_tensor(dec['input_ids']).reshape(-1)
        labels = torch.roll(dec_ids, shifts=-1, dims=0)
        labels[-1] = -100
        labels[labels == self.tokenizer.pad_token_id] = -100
        vis, roi, coors = self.producer.visual(s['original_idx'], photos)
        return (vis, roi, coors, labels, dec_ids, enc_ids, enc_type, enc_mask, added, s['target_aspect'], text)
