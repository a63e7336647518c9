"""Drop-in surface of fcmf_framework (CPU: construction, names, state-dict keys, error behaviour).
Mirrors what the reference's drivers rely on."""
import pytest
import torch

from helpers import make_hf_dir
import synthetic_data as synth


@pytest.fixture(scope="module")
def tiny_dir():
    return make_hf_dir(synth.TINY_CFG)


def test_star_exports_and_constants():
    import fcmf_framework.mm_modeling as mm
    for name in ("HIDDEN_SIZE", "NUM_HIDDEN_LAYERS", "NUM_ATTENTION_HEADS", "INTERMEDIATE_SIZE", "HIDDEN_ACT",
                 "HIDDEN_DROPOUT_PROB", "ATTENTION_PROBS_DROPOUT_PROB", "MAX_POSITION_EMBEDDINGS", "ACT2FN", "gelu",
                 "Attention", "AttentionPooler", "FCMFLayerNorm", "BertSelfAttention", "BertCoAttention",
                 "BertSelfOutput", "BertAttention", "BertCrossAttention", "BertIntermediate", "BertOutput",
                 "BertLayer", "BertCrossAttentionLayer", "MultimodalEncoder", "BertCrossEncoder", "BertText1Pooler",
                 "BertPooler", "FeatureExtractor", "MultimodalDenoisingEncoder", "PositionWiseFFN", "AddNorm",
                 "TransformerDecoderBlock", "PositionalEncoding", "IAOGDecoder"):
        assert hasattr(mm, name), name
    assert (mm.HIDDEN_SIZE, mm.NUM_ATTENTION_HEADS, mm.INTERMEDIATE_SIZE) == (768, 12, 3072)
    import fcmf_framework.roi_modeling as rm
    assert all(hasattr(rm, n) for n in ("clones", "box_attention", "BoxMultiHeadedAttention"))
    from fcmf_framework.resnet_utils import myResNetImg, myResNetRoI  # noqa: F401
    from fcmf_framework.optimization import BertAdam, SCHEDULES, warmup_linear  # noqa: F401
    import inspect
    from fcmf_framework.fcmf_pretraining import beam_search        # (fcmf_pretraining.py:383-386: the reference's signature)
    assert list(inspect.signature(beam_search).parameters) == [
        "model", "tokenizer", "enc_ids", "enc_mask", "enc_type", "add_mask", "vis_embeds", "roi_embeds", "roi_coors",
        "beam_size", "num_preds", "max_len", "device"]
    assert set(SCHEDULES) == {"warmup_cosine", "warmup_constant", "warmup_linear"}


def test_fcmf_state_dict_keys_and_param_groups(tiny_dir):
    from fcmf_framework.fcmf_multimodal import FCMF
    from oracle import fcmf_oracle as O
    m = FCMF(tiny_dir, num_labels=4, num_imgs=2, num_roi=5)
    exp = synth.fcmf_param_shapes(synth.TINY_CFG)
    sd = m.state_dict()
    assert list(sd.keys()) == list(sd.keys()) and set(sd) == set(exp)
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in exp.items())
    # attribute tree the drivers touch (run_multimodal_fcmf.py:223,232-233,249-262)
    assert m.encoder.bert.cell.resize_token_embeddings(600).num_embeddings == 600
    assert m.state_dict()["encoder.bert.cell.embeddings.word_embeddings.weight"].shape[0] == 600
    for attr in ("text_pooler", "dropout", "classifier"):
        assert hasattr(m, attr)
    groups = O.fcmf_param_groups([n for n, _ in m.named_parameters()])
    assert "classifier.weight" in groups[2]["names"] and "text_pooler.dense.bias" in groups[3]["names"]
    assert "encoder.mm_attention.layer.0.output.LayerNorm.weight" in groups[1]["names"]
    # IAOG -> fine-tune transfer: encoder.* keys load with strict=False (run_multimodal_fcmf.py:385-391)
    enc_sd = {k: v for k, v in m.state_dict().items() if k.startswith("encoder.")}
    missing, unexpected = m.load_state_dict(enc_sd, strict=False)
    assert not unexpected and all(k.startswith(("text_pooler", "classifier")) for k in missing)


def test_seq2seq_tying_and_keys(tiny_dir):
    from fcmf_framework.fcmf_pretraining import FCMFSeq2Seq
    m = FCMFSeq2Seq(512, 20, tiny_dir, 2, 5, 1.0)
    we = m.encoder.bert.cell.embeddings.word_embeddings.weight
    assert m.decoder.embedding.weight is we and m.decoder.dense.weight is we
    m.decoder.embedding = torch.nn.Embedding(512, m.decoder.num_hiddens)     # run_pretraining_fcmf.py:189
    assert m.decoder.dense.weight is we and m.decoder.embedding.weight is not we
    sd = m.state_dict()
    exp = synth.iaog_decoder_param_shapes(synth.TINY_CFG, 512)
    assert all(k in sd and tuple(sd[k].shape) == tuple(v) for k, v in exp.items())
    assert tuple(sd["decoder.pos_encoding.P"].shape) == (1, 512, 64)
    assert m.decoder.num_blks == 2 and len(m.decoder.init_state(None, None)[2]) == 2
    # Linear / Embedding weights of the fusion blocks were re-drawn N(0, 0.02); custom LayerNorms untouched
    assert abs(m.encoder.vismap2text.weight.std().item() - 0.02) < 2e-3
    assert torch.all(m.decoder.blks.block0.addnorm1.ln.weight == 1)


def test_error_behaviour_matches_reference():
    from fcmf_framework.optimization import BertAdam
    p = [torch.nn.Parameter(torch.zeros(3))]
    with pytest.raises(ValueError):
        BertAdam(p, lr=-1.0)
    with pytest.raises(ValueError):
        BertAdam(p, lr=1e-3, schedule="nope")
    with pytest.raises(ValueError):
        BertAdam(p, lr=1e-3, warmup=1.5)
    with pytest.raises(ValueError):
        BertAdam(p, lr=1e-3, b1=1.0)
    assert BertAdam(p, lr=1e-3).get_lr() == [0]
    from fcmf_framework.mm_modeling import Attention
    with pytest.raises(RuntimeError):
        Attention(64, 16, 4, score_function="nonsense")


def test_cpu_tensors_fail_loudly(tiny_dir):
    """the product has no CPU fallback: a CPU call must raise, not silently compute"""
    from fcmf_framework.fcmf_multimodal import FCMF
    from fcmf_framework._hip import HipLibraryError
    m = FCMF(tiny_dir, num_imgs=2, num_roi=5)
    with pytest.raises(HipLibraryError):
        m(torch.zeros(2, 16, dtype=torch.long), torch.zeros(2, 2, 49, 2048), torch.zeros(2, 2, 5, 2048),
          torch.zeros(2, 2, 5, 4), None, torch.ones(2, 16, dtype=torch.long), torch.ones(2, 65, dtype=torch.long))


def test_product_never_imports_oracle():
    import os
    from conftest import PKG
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


# The two command lines the reference publishes (argument lists only; paths as printed in the notebooks):
#   Experimental_Results/Pretraining/Pretraining-Notebook.ipynb:6927-6945  and
#   Experimental_Results/ViIM_FCMF/ViIM-FCMF-notebook.ipynb:805-825
PUBLISHED_PRETRAIN = """--data_dir /kaggle/input/vimacsa/ViMACSA --pretrained_data_dir /kaggle/input/iaog-filtered
 --list_aspect Location Food Room Facilities Service Public_area --num_imgs 7 --num_rois 4
 --image_dir /kaggle/input/vimacsa/ViMACSA/image --pretrained_hf_model /kaggle/input/uitnlpvisobert/pytorch/default/1
 --output_dir /kaggle/working/ViMACSA/output/pretraining --train_batch_size 16 --eval_batch_size 64 --num_train_epochs 30
 --learning_rate 3e-5 --warmup_proportion 0.1 --alpha 1.0 --gradient_accumulation_steps 1 --max_len_decoder 8 --do_train --fp16""".split()
PUBLISHED_FINETUNE = """--data_dir /kaggle/input/implicit-vimacsa --output_dir /kaggle/working/ViMACSA/output_iaog_fcmf
 --image_dir /kaggle/input/vimacsa/ViMACSA/image --pretrained_hf_model /kaggle/input/uitnlpvisobert/pytorch/default/1
 --pretrained_iaog_path /kaggle/input/iaog-last-checkpoint/pytorch/30_epoch/3/seed_42_iaog_model_last.pth
 --list_aspect Location Food Room Facilities Service Public_area --num_polarity 4 --num_imgs 7 --num_rois 4
 --train_batch_size 4 --eval_batch_size 32 --num_train_epochs 13 --encoder_learning_rate 7e-5
 --classifier_head_learning_rate 7e-4 --warmup_proportion 0.1 --gradient_accumulation_steps 2 --do_train --do_eval --fp16
 --alpha 1""".split()
# every flag the reference's two parsers declare (run_pretraining_fcmf.py:45-84, run_multimodal_fcmf.py:60-110) with a value of
# the declared type: the drop-in drivers must accept all of them
REF_PRETRAIN_FLAGS = {"--data_dir": "d", "--pretrained_data_dir": "p", "--output_dir": "o", "--pretrained_hf_model": "m",
                      "--bert_score_model": "uitnlp/visobert", "--image_dir": "i", "--resnet_label_path": "r",
                      "--resume_from_checkpoint": "c", "--max_seq_length": "170", "--max_len_decoder": "20", "--num_imgs": "7",
                      "--num_rois": "4", "--fine_tune_cnn": None, "--alpha": "0.8", "--beam_size": "2", "--do_train": None,
                      "--do_eval": None, "--train_batch_size": "8", "--eval_batch_size": "8", "--learning_rate": "3e-5",
                      "--num_train_epochs": "8.0", "--warmup_proportion": "0.1", "--gradient_accumulation_steps": "1",
                      "--seed": "42", "--fp16": None, "--adam_epsilon": "1e-8", "--no_cuda": None, "--ddp": None,
                      "--list_aspect": "Location"}


def _argv(flags):
    out = []
    for k, v in flags.items():
        out += [k] if v is None else [k, v]
    return out


def test_drivers_accept_the_published_command_lines():
    import run_multimodal_fcmf as ft
    import run_pretraining_fcmf as pt
    a = pt.build_parser().parse_args(PUBLISHED_PRETRAIN)
    assert a.list_aspect == ['Location', 'Food', 'Room', 'Facilities', 'Service', 'Public_area']
    assert (a.max_len_decoder, a.train_batch_size, a.alpha, a.fp16, a.do_train) == (8, 16, 1.0, True, True)
    # defaults of the flags that command line leaves out are the reference's (run_pretraining_fcmf.py:53-66)
    assert (a.max_seq_length, a.beam_size, a.bert_score_model, a.resnet_label_path) == (
        170, 2, 'uitnlp/visobert', '/kaggle/input/resnet-output')
    b = ft.build_parser().parse_args(PUBLISHED_FINETUNE)
    assert b.list_aspect[-1] == 'Public_area' and b.num_polarity == 4 and b.gradient_accumulation_steps == 2
    assert (b.encoder_learning_rate, b.classifier_head_learning_rate, b.alpha) == (7e-5, 7e-4, 1.0)
    c = pt.build_parser().parse_args(_argv(REF_PRETRAIN_FLAGS))
    assert c.list_aspect == ['Location'] and c.beam_size == 2 and c.fine_tune_cnn and c.ddp
