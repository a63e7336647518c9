"""RoBERTa-family text encoder on the HIP kernels.

The reference loads `transformers.AutoModel.from_pretrained(path, attn_implementation="eager")`
(mm_modeling.py:436-438) -- third-party code.  This module restates that model (HF
modeling_roberta.py: RobertaEmbeddings, RobertaLayer, RobertaPooler) with the SAME attribute
tree and state-dict keys (`embeddings.word_embeddings.weight`, `encoder.layer.N.attention.self.
query.weight`, ...), reads the same `config.json` + `model.safetensors` / `pytorch_model.bin`
directory, and runs every layer on libfcmf_hip.so.  nn.Linear / nn.Embedding / nn.LayerNorm are
used as parameter containers only (so isinstance-based init code such as
fcmf_pretraining.py:209-221 keeps working); their torch forward is never called.
"""
import json
import os

import torch
import torch.nn as nn

from . import layers, ops
from .fused import QKVStorageMixin


class RobertaConfig:
    def __init__(self, **kw):
        self.vocab_size = kw.get("vocab_size", 64001)
        self.hidden_size = kw.get("hidden_size", 768)
        self.num_hidden_layers = kw.get("num_hidden_layers", 12)
        self.num_attention_heads = kw.get("num_attention_heads", 12)
        self.intermediate_size = kw.get("intermediate_size", 3072)
        self.hidden_act = kw.get("hidden_act", "gelu")
        self.hidden_dropout_prob = kw.get("hidden_dropout_prob", 0.1)
        self.attention_probs_dropout_prob = kw.get("attention_probs_dropout_prob", 0.1)
        self.max_position_embeddings = kw.get("max_position_embeddings", 258)
        self.type_vocab_size = kw.get("type_vocab_size", 1)
        self.layer_norm_eps = kw.get("layer_norm_eps", 1e-5)
        self.pad_token_id = kw.get("pad_token_id", 1)
        self.initializer_range = kw.get("initializer_range", 0.02)
        if self.hidden_act != "gelu":
            raise ValueError("only the erf-GELU text encoder is implemented (hidden_act='gelu')")

    @classmethod
    def from_pretrained(cls, path):
        with open(os.path.join(path, "config.json")) as f:
            return cls(**json.load(f))

    def to_dict(self):
        return dict(self.__dict__)


class RobertaEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.word_embeddings = nn.Embedding(cfg.vocab_size, cfg.hidden_size, padding_idx=cfg.pad_token_id)
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size, padding_idx=cfg.pad_token_id)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.dropout = nn.Dropout(cfg.hidden_dropout_prob)
        self.padding_idx = cfg.pad_token_id

    def forward(self, input_ids, token_type_ids=None):
        pos = ops.position_ids(input_ids, self.padding_idx)
        return ops.embed_layer_norm(input_ids, pos, token_type_ids, self.word_embeddings.weight,
                                    self.position_embeddings.weight, self.token_type_embeddings.weight,
                                    self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps,
                                    self.dropout.p, self.training, self.padding_idx, ops.compute_dtype())


class _SelfAttention(QKVStorageMixin, nn.Module):
    def __init__(self, H, heads, p):
        super().__init__()
        self.num_attention_heads = heads
        self.attention_head_size = H // heads
        self.all_head_size = H
        self.query = nn.Linear(H, H)
        self.key = nn.Linear(H, H)
        self.value = nn.Linear(H, H)
        self.dropout = nn.Dropout(p)
        self._fuse_qkv_storage()


class _SelfOutput(nn.Module):
    def __init__(self, H, eps, p):
        super().__init__()
        self.dense = nn.Linear(H, H)
        self.LayerNorm = nn.LayerNorm(H, eps=eps)
        self.dropout = nn.Dropout(p)


class _Attention(nn.Module):
    def __init__(self, H, heads, eps, p_hidden, p_attn):
        super().__init__()
        self.self = _SelfAttention(H, heads, p_attn)
        self.output = _SelfOutput(H, eps, p_hidden)


class _Intermediate(nn.Module):
    def __init__(self, H, I):
        super().__init__()
        self.dense = nn.Linear(H, I)


class _Output(nn.Module):
    def __init__(self, H, I, eps, p):
        super().__init__()
        self.dense = nn.Linear(I, H)
        self.LayerNorm = nn.LayerNorm(H, eps=eps)
        self.dropout = nn.Dropout(p)


class RobertaLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        H, I = cfg.hidden_size, cfg.intermediate_size
        self.attention = _Attention(H, cfg.num_attention_heads, cfg.layer_norm_eps, cfg.hidden_dropout_prob,
                                    cfg.attention_probs_dropout_prob)
        self.intermediate = _Intermediate(H, I)
        self.output = _Output(H, I, cfg.layer_norm_eps, cfg.hidden_dropout_prob)
        self.heads = cfg.num_attention_heads
        self.eps = cfg.layer_norm_eps

    def forward(self, hidden_states, add_mask):
        return layers.transformer_layer(self, hidden_states, hidden_states, add_mask, self.heads, self.eps,
                                        self.output.dropout.p, self.attention.self.dropout.p, self.training)


class RobertaEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([RobertaLayer(cfg) for _ in range(cfg.num_hidden_layers)])

    def forward(self, x, add_mask):
        for l in self.layer:
            x = l(x, add_mask)
        return x


class RobertaPooler(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)

    def forward(self, x):
        return ops.linear(x[:, 0], self.dense.weight, self.dense.bias, act="tanh")


class RobertaModel(nn.Module):
    """Drop-in for the HF model held in FeatureExtractor.cell (mm_modeling.py:436)."""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.embeddings = RobertaEmbeddings(cfg)
        self.encoder = RobertaEncoder(cfg)
        self.pooler = RobertaPooler(cfg)
        std = cfg.initializer_range
        for m in self.modules():  # HF _init_weights
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(0.0, std)
                m.bias.data.zero_()
            elif isinstance(m, nn.Embedding):
                m.weight.data.normal_(0.0, std)
                if m.padding_idx is not None:
                    m.weight.data[m.padding_idx].zero_()

    # ---- loading -----------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path, **unused):
        """Reads an HF model directory without executing anything from it: config.json plus
        model.safetensors (preferred) or pytorch_model.bin (torch.load(weights_only=True))."""
        if not os.path.isdir(path):
            raise OSError(f"{path} is not a local model directory (local_files_only semantics, "
                          f"mm_modeling.py:436-438)")
        cfg = RobertaConfig.from_pretrained(path)
        model = cls(cfg)
        st = os.path.join(path, "model.safetensors")
        pb = os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        elif os.path.exists(pb):
            sd = torch.load(pb, map_location="cpu", weights_only=True)
        else:
            raise OSError(f"no model.safetensors / pytorch_model.bin under {path}")
        clean = {}
        for k, v in sd.items():
            for pre in ("roberta.", "bert.", "model."):
                if k.startswith(pre):
                    k = k[len(pre):]
            if k.startswith("embeddings.position_ids") or k.startswith("embeddings.token_type_ids") or k.startswith("lm_head"):
                continue
            clean[k] = v.float()
        missing, unexpected = model.load_state_dict(clean, strict=False)
        missing = [m for m in missing if not m.startswith("pooler.")]
        if missing:
            raise RuntimeError(f"checkpoint under {path} lacks {missing[:4]}...")
        return model

    def save_pretrained(self, path):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(dict(self.config.to_dict(), model_type="roberta", architectures=["RobertaModel"]), f)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "model.safetensors"))

    def resize_token_embeddings(self, new_num_tokens=None):
        """HF PreTrainedModel.resize_token_embeddings (run_multimodal_fcmf.py:223): keep the first
        min(old,new) rows, N(0, initializer_range) for the new ones."""
        old = self.embeddings.word_embeddings
        if new_num_tokens is None or new_num_tokens == old.num_embeddings:
            return old
        new = nn.Embedding(new_num_tokens, old.embedding_dim, padding_idx=old.padding_idx)
        new.to(old.weight.device, old.weight.dtype)
        new.weight.data.normal_(0.0, self.config.initializer_range)
        n = min(old.num_embeddings, new_num_tokens)
        new.weight.data[:n] = old.weight.data[:n]
        self.embeddings.word_embeddings = new
        self.config.vocab_size = new_num_tokens
        return new

    # ---- forward -----------------------------------------------------------------------
    def encode(self, input_ids, token_type_ids=None, attention_mask=None):
        """sequence output [B,S,H] only (the training path never uses pooled/attentions)"""
        x = self.embeddings(input_ids, token_type_ids)
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        add_mask = layers.additive_mask(attention_mask, input_ids.shape[1], torch.finfo(torch.float32).min)
        return self.encoder(x, add_mask)

    def forward(self, input_ids=None, token_type_ids=None, attention_mask=None, output_attentions=False, **unused):
        seq = self.encode(input_ids, token_type_ids, attention_mask)
        pooled = self.pooler(seq)
        # attention probabilities are never materialised by the fused kernels
        return (seq, pooled, ())
