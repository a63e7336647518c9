"""Deterministic synthetic weights and batches (data generator only: no model arithmetic).

No dataset, tokenizer or checkpoint is reachable offline, so both the
golden fixtures and the benchmark use seeded synthetic data of the reference's shapes.
Weights come from NumPy's PCG64 stream seeded by crc32(parameter name) -- NumPy guarantees
that stream is stable across versions/platforms -- so no weight file has to be committed:
the fixture script (which imports the reference) and the tests (which do not) regenerate
bit-identical weights.

Shapes follow SURVEY.md Appendix A (state-dict keys verified against the reference import).
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

TINY_CFG = dict(vocab_size=512, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                intermediate_size=128, max_position_embeddings=40, type_vocab_size=1,
                pad_token_id=1, layer_norm_eps=1e-5, hidden_dropout_prob=0.1,
                attention_probs_dropout_prob=0.1)
BASE_CFG = dict(vocab_size=64001, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                intermediate_size=3072, max_position_embeddings=258, type_vocab_size=1,
                pad_token_id=1, layer_norm_eps=1e-5, hidden_dropout_prob=0.1,
                attention_probs_dropout_prob=0.1)


# FCMF-large (BASELINE.json configs[4]): XLM-R-large geometry; the vocabulary is kept at 64001 rows (the embedding table
# is a gather, its row count does not change any kernel shape) so that the test model stays ~0.4 G parameters
LARGE_CFG = dict(vocab_size=64001, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                 intermediate_size=4096, max_position_embeddings=514, type_vocab_size=1,
                 pad_token_id=1, layer_norm_eps=1e-5, hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1)


def _layer_shapes(prefix, H, I, out):
    for n in ("query", "key", "value"):
        out[f"{prefix}.attention.self.{n}.weight"] = (H, H)
        out[f"{prefix}.attention.self.{n}.bias"] = (H,)
    out[f"{prefix}.attention.output.dense.weight"] = (H, H)
    out[f"{prefix}.attention.output.dense.bias"] = (H,)
    out[f"{prefix}.attention.output.LayerNorm.weight"] = (H,)
    out[f"{prefix}.attention.output.LayerNorm.bias"] = (H,)
    out[f"{prefix}.intermediate.dense.weight"] = (I, H)
    out[f"{prefix}.intermediate.dense.bias"] = (I,)
    out[f"{prefix}.output.dense.weight"] = (H, I)
    out[f"{prefix}.output.dense.bias"] = (H,)
    out[f"{prefix}.output.LayerNorm.weight"] = (H,)
    out[f"{prefix}.output.LayerNorm.bias"] = (H,)


def roberta_param_shapes(cfg, prefix="", pooler=True):
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    out = OrderedDict()
    out[f"{prefix}embeddings.word_embeddings.weight"] = (cfg["vocab_size"], H)
    out[f"{prefix}embeddings.position_embeddings.weight"] = (cfg["max_position_embeddings"], H)
    out[f"{prefix}embeddings.token_type_embeddings.weight"] = (cfg["type_vocab_size"], H)
    out[f"{prefix}embeddings.LayerNorm.weight"] = (H,)
    out[f"{prefix}embeddings.LayerNorm.bias"] = (H,)
    for l in range(cfg["num_hidden_layers"]):
        _layer_shapes(f"{prefix}encoder.layer.{l}", H, I, out)
    if pooler:
        out[f"{prefix}pooler.dense.weight"] = (H, H)
        out[f"{prefix}pooler.dense.bias"] = (H,)
    return out


def fcmf_param_shapes(cfg, num_labels=4, feat_dim=2048):
    """FCMF (fine-tune model) state-dict keys -> shapes, SURVEY.md Appendix A."""
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    out = roberta_param_shapes(cfg, "encoder.bert.cell.")
    for n in ("vismap2text", "roimap2text"):
        out[f"encoder.{n}.weight"] = (H, feat_dim)
        out[f"encoder.{n}.bias"] = (H,)
    for i in range(4):
        out[f"encoder.box_head.linears.{i}.weight"] = (H, H)
        out[f"encoder.box_head.linears.{i}.bias"] = (H,)
    for i in range(8):
        out[f"encoder.box_head.WGs.{i}.weight"] = (1, 64)
        out[f"encoder.box_head.WGs.{i}.bias"] = (1,)
    _layer_shapes("encoder.text2img_attention.layer.0", H, I, out)
    for n in ("text2img_pooler", "text2roi_pooler"):
        out[f"encoder.{n}.dense.weight"] = (H, H)
        out[f"encoder.{n}.dense.bias"] = (H,)
    _layer_shapes("encoder.mm_attention.layer.0", H, I, out)
    out["text_pooler.dense.weight"] = (H, H)
    out["text_pooler.dense.bias"] = (H,)
    out["classifier.weight"] = (num_labels, H)
    out["classifier.bias"] = (num_labels,)
    return out


def iaog_decoder_param_shapes(cfg, vocab):
    H, nh = cfg["hidden_size"], cfg["num_attention_heads"]
    d = H // nh
    out = OrderedDict()
    out["decoder.embedding.weight"] = (vocab, H)
    for i in range(cfg["num_hidden_layers"]):
        b = f"decoder.blks.block{i}"
        for a in ("attention1", "attention2"):
            out[f"{b}.{a}.w_kx"] = (nh, H, d)
            out[f"{b}.{a}.w_qx"] = (nh, H, d)
            out[f"{b}.{a}.proj.weight"] = (H, H)
            out[f"{b}.{a}.proj.bias"] = (H,)
        for ln in ("addnorm1", "addnorm2", "add_norm3"):
            out[f"{b}.{ln}.ln.weight"] = (H,)
            out[f"{b}.{ln}.ln.bias"] = (H,)
        out[f"{b}.ffn.dense1.weight"] = (H, H)
        out[f"{b}.ffn.dense1.bias"] = (H,)
        out[f"{b}.ffn.dense2.weight"] = (H, H)
        out[f"{b}.ffn.dense2.bias"] = (H,)
    out["decoder.dense.bias"] = (vocab,)
    return out


def synth_param(name, shape, seed=0):
    """One parameter tensor: N(0, 0.02) for matrices/embeddings/biases (biases non-zero on
    purpose so that bias paths are exercised), 1 + N(0, 0.02) for LayerNorm/ln weights;
    WGs weights are N(0, 0.1) + 0.05 so that relu(WG(emb)) is neither all-dead nor all-live."""
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1 & 0xFFFFFFFF)))
    a = rng.standard_normal(size=shape, dtype=np.float32)
    if name.endswith("LayerNorm.weight") or name.endswith(".ln.weight"):
        a = 1.0 + 0.02 * a
    elif ".WGs." in name:
        a = 0.1 * a + 0.05
    else:
        a = 0.02 * a
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def synth_params(shapes, seed=0):
    return OrderedDict((n, synth_param(n, s, seed)) for n, s in shapes.items())


def synth_batch(B, cfg, S=128, num_imgs=7, num_roi=36, num_aspects=6, num_labels=4,
                feat_dim=2048, seed=42, min_len=None, coord_dtype=torch.float64):
    """Synthetic fine-tune batch, SURVEY.md section 8(d): <s>=0 first, length l~U{min_len..S},
    tokens U{3..V-1}, </s>=2 at l-1, pad=1 after; added mask all ones; non-negative features;
    k~U{0..num_roi} valid ROIs per image (others all-zero rows and zero boxes, mirroring
    vimacsa_dataset.py:169-172); boxes (x1,x2,y1,y2) with sorted pairs in [0,1], float64."""
    rng = np.random.Generator(np.random.PCG64(seed))
    V = cfg["vocab_size"]
    min_len = min_len if min_len is not None else max(4, S // 4)
    ids = np.full((B, num_aspects, S), cfg["pad_token_id"], dtype=np.int64)
    for b in range(B):
        for a in range(num_aspects):
            l = int(rng.integers(min_len, S + 1))
            ids[b, a, 0] = 0
            ids[b, a, 1:l - 1] = rng.integers(3, V, size=l - 2)
            ids[b, a, l - 1] = 2
    att = (ids != cfg["pad_token_id"]).astype(np.int64)
    added = np.ones((B, num_aspects, S + max(49, num_roi)), dtype=np.int64)   # covers text + patches and text + ROIs
    vis = np.abs(rng.standard_normal((B, num_imgs, 49, feat_dim), dtype=np.float32)) * 0.5
    roi = np.abs(rng.standard_normal((B, num_imgs, num_roi, feat_dim), dtype=np.float32)) * 0.5
    xs = np.sort(rng.random((B, num_imgs, num_roi, 2)), axis=-1)
    ys = np.sort(rng.random((B, num_imgs, num_roi, 2)), axis=-1)
    coors = np.concatenate([xs, ys], axis=-1)
    kvalid = rng.integers(0, num_roi + 1, size=(B, num_imgs))
    valid = (np.arange(num_roi)[None, None, :] < kvalid[..., None])
    roi = roi * valid[..., None]
    coors = coors * valid[..., None]
    labels = rng.integers(0, num_labels, size=(B, num_aspects))
    return dict(
        input_ids=torch.from_numpy(ids), token_type_ids=torch.zeros(B, num_aspects, S, dtype=torch.long),
        attention_mask=torch.from_numpy(att), added_attention_mask=torch.from_numpy(added),
        visual_embeds_att=torch.from_numpy(vis), roi_embeds_att=torch.from_numpy(roi.astype(np.float32)),
        roi_coors=torch.from_numpy(coors).to(coord_dtype), labels=torch.from_numpy(labels))


# ---------------------------------------------------------------------------------------
# ResNet-152 trunk (torchvision layout: v1.5 bottleneck, stride on the 3x3; resnet_utils.py:13-24 consumes
# conv1/bn1/relu/maxpool/layer1..4).  torchvision is absent offline: shapes follow its published state-dict.
# ---------------------------------------------------------------------------------------
RESNET152_LAYERS = (3, 8, 36, 3)
RESNET_TINY_LAYERS = (1, 2, 2, 1)      # same block type, fewer blocks: test-size trunk


def resnet_param_shapes(layers=RESNET152_LAYERS, base=64, with_fc=False):
    """state-dict keys -> shapes of a torchvision Bottleneck ResNet (parameters AND BatchNorm buffers)"""
    out = OrderedDict()

    def bn(prefix, c):
        out[f"{prefix}.weight"] = (c,)
        out[f"{prefix}.bias"] = (c,)
        out[f"{prefix}.running_mean"] = (c,)
        out[f"{prefix}.running_var"] = (c,)
        out[f"{prefix}.num_batches_tracked"] = ()
    out["conv1.weight"] = (base, 3, 7, 7)
    bn("bn1", base)
    inplanes = base
    for li, nblk in enumerate(layers):
        planes = base * (2 ** li)
        for b in range(nblk):
            p = f"layer{li + 1}.{b}"
            out[f"{p}.conv1.weight"] = (planes, inplanes, 1, 1)
            bn(f"{p}.bn1", planes)
            out[f"{p}.conv2.weight"] = (planes, planes, 3, 3)
            bn(f"{p}.bn2", planes)
            out[f"{p}.conv3.weight"] = (planes * 4, planes, 1, 1)
            bn(f"{p}.bn3", planes * 4)
            if b == 0:
                out[f"{p}.downsample.0.weight"] = (planes * 4, inplanes, 1, 1)
                bn(f"{p}.downsample.1", planes * 4)
            inplanes = planes * 4
    if with_fc:
        out["fc.weight"] = (1000, inplanes)
        out["fc.bias"] = (1000,)
    return out


def synth_resnet_params(shapes, seed=0):
    """Kaiming-normal (fan_out) conv weights as torchvision initialises them; BatchNorm weight 1 + N(0, 0.02)
    (0.25 x that on the last BN of a block so that 50 residual additions stay tame in eval mode), bias N(0, 0.02),
    running_mean N(0, 0.1), running_var U(0.5, 1.5), num_batches_tracked 0."""
    out = OrderedDict()
    for name, shape in shapes.items():
        rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1 & 0xFFFFFFFF)))
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros((), dtype=torch.long)
            continue
        if len(shape) == 4:
            fan_out = shape[0] * shape[2] * shape[3]
            a = rng.standard_normal(size=shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_out))
        elif name.endswith("running_var"):
            a = (0.5 + rng.random(size=shape, dtype=np.float32)).astype(np.float32)
        elif name.endswith("running_mean"):
            a = 0.1 * rng.standard_normal(size=shape, dtype=np.float32)
        elif name.endswith(".weight") and len(shape) == 1:
            a = 1.0 + 0.02 * rng.standard_normal(size=shape, dtype=np.float32)
            if name.endswith("bn3.weight"):
                a = 0.25 * a
        else:
            a = 0.02 * rng.standard_normal(size=shape, dtype=np.float32)
        out[name] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return out


def synth_crops(n, size=224, seed=0):
    """[n, 3, size, size] float32 'normalised image' crops (what image_process.py hands the trunk)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal((n, 3, size, size), dtype=np.float32))
