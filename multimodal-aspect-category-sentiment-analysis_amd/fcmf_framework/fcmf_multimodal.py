"""FCMF fine-tune model: encoder -> [CLS] pooler -> dropout -> Linear(H, num_labels).

Same surface as the reference's fcmf_framework/fcmf_multimodal.py:12-51 (ctor, forward keywords,
.encoder/.text_pooler/.dropout/.classifier, state-dict keys).  `forward_aspects` is the
MI355X-first entry the training driver uses: all aspects of a batch in one pass
(run_multimodal_fcmf.py:463-475 runs 6 separate forwards instead).
"""
import torch
import torch.nn as nn

from . import ops
from .mm_modeling import *  # noqa: F401,F403
from .mm_modeling import BertPooler, HIDDEN_DROPOUT_PROB
from .roi_modeling import *  # noqa: F401,F403
from .fcmf_pretraining import FCMFEncoder


class FCMF(nn.Module):
    def __init__(self, pretrained_path, num_labels=4, num_imgs=7, num_roi=7, alpha=0.7):
        super().__init__()
        self.encoder = FCMFEncoder(pretrained_path, num_imgs, num_roi, alpha)
        H = self.encoder.bert.cell.config.hidden_size
        self.text_pooler = BertPooler(H)
        self.dropout = nn.Dropout(HIDDEN_DROPOUT_PROB)
        self.classifier = nn.Linear(H, num_labels)

    def _init_weights(self, module):
        """BERT-style init (defined but not applied by the reference, fcmf_multimodal.py:19-38)"""
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    def apply_custom_init(self, module):
        module.apply(self._init_weights)

    def _head(self, sequence_output):
        cls_output = self.text_pooler(sequence_output)
        pooled_output = ops.dropout(cls_output, self.dropout.p, self.training)
        return ops.linear(pooled_output, self.classifier.weight, self.classifier.bias)

    def forward(self, input_ids, visual_embeds_att, roi_embeds_att, roi_coors=None, token_type_ids=None,
                attention_mask=None, added_attention_mask=None):
        output = self.encoder(input_ids, visual_embeds_att, roi_embeds_att, roi_coors, token_type_ids,
                              attention_mask, added_attention_mask)
        sequence_output = output[0] if isinstance(output, tuple) else output
        return self._head(sequence_output).float()

    def forward_aspects(self, input_ids, visual_embeds_att, roi_embeds_att, roi_coors=None, token_type_ids=None,
                        attention_mask=None, added_attention_mask=None):
        """All aspects at once: input_ids/token_type_ids/attention_mask [B,A,S], added mask [B,A,L]
        -> logits [B,A,num_labels]; equals stacking `forward` over the aspect axis."""
        B, A, _ = input_ids.shape
        seq = self.encoder.encode_aspects(input_ids, visual_embeds_att, roi_embeds_att, roi_coors, token_type_ids,
                                          attention_mask, added_attention_mask)
        return self._head(seq).float().view(B, A, -1)

    def loss_aspects(self, logits, labels):
        """sum over aspects of the batch-mean CE (run_multimodal_fcmf.py:463-475), in one kernel:
        mean over B*A rows times A."""
        B, A, C = logits.shape
        return ops.cross_entropy(logits.reshape(B * A, C), labels.reshape(B * A), mult=float(A))
