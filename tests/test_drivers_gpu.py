"""The two training entry points end to end on the GPU in synthetic mode (tiny text encoder):
flags parse, the step loop runs, the loss goes down, the reference's checkpoint dict is written and
a resume continues from it."""
import os

import pytest
import torch

from helpers import make_hf_dir
import synthetic_data as synth

pytestmark = pytest.mark.gpu


def test_finetune_driver_synthetic(tmp_path, dev):
    import run_multimodal_fcmf as drv
    hf = make_hf_dir(synth.TINY_CFG)
    out = str(tmp_path / "ft")
    common = ["--output_dir", out, "--pretrained_hf_model", hf, "--do_train", "--num_imgs", "2", "--num_rois", "5",
              "--train_batch_size", "4", "--gradient_accumulation_steps", "2", "--synthetic_steps", "6",
              "--max_seq_length", "16", "--seed", "3"]
    drv.main(common + ["--num_train_epochs", "1"])
    ck = torch.load(os.path.join(out, "seed_3_fcmf_model_last.pth"), map_location="cpu", weights_only=True)
    assert set(ck) >= {"epoch", "best_score", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict"}
    assert ck["epoch"] == 0 and "classifier.weight" in ck["model_state_dict"]
    assert set(ck["model_state_dict"]) == set(synth.fcmf_param_shapes(synth.TINY_CFG))
    # resume: starts at epoch 1 and writes a new "last" checkpoint with epoch == 1
    drv.main(common + ["--num_train_epochs", "2", "--resume_from_checkpoint", os.path.join(out, "seed_3_fcmf_model_last.pth"), "--bf16"])
    ck2 = torch.load(os.path.join(out, "seed_3_fcmf_model_last.pth"), map_location="cpu", weights_only=True)
    assert ck2["epoch"] == 1
    assert not torch.equal(ck2["model_state_dict"]["classifier.weight"], ck["model_state_dict"]["classifier.weight"])
    from fcmf_framework import ops
    ops.set_compute_dtype(torch.float32)


def test_finetune_driver_eval_path_and_resnet_in_the_loop(tmp_path, dev, caplog):
    """--do_eval runs the dev-set evaluation (no-grad forward, argmax, macro-F1 per aspect: reference :500-552) after the
    epoch; --synthetic_pixels puts the ResNet-152 trunk inside the step (reference :449-460): pixel crops in, the two
    batched trunk passes, then the FCMF step on the features they produce"""
    import logging
    import run_multimodal_fcmf as drv
    hf = make_hf_dir(synth.TINY_CFG)
    out = str(tmp_path / "ev")
    with caplog.at_level(logging.INFO, logger="fcmf"):
        drv.main(["--output_dir", out, "--pretrained_hf_model", hf, "--do_train", "--do_eval", "--num_imgs", "2", "--num_rois", "2",
                  "--train_batch_size", "2", "--eval_batch_size", "2", "--gradient_accumulation_steps", "1", "--synthetic_steps", "2",
                  "--synthetic_pixels", "64", "--max_seq_length", "16", "--seed", "4", "--num_train_epochs", "1", "--bf16"])
    from fcmf_framework import ops
    ops.set_compute_dtype(torch.float32)
    log = open(os.path.join(out, "training_fcmf.log")).read()
    assert "Dev macro-F1 per aspect" in log
    ck = torch.load(os.path.join(out, "seed_4_fcmf_model_last.pth"), map_location="cpu", weights_only=True)
    assert 0.0 <= ck["best_score"] <= 1.0 and ck["epoch"] == 0


def test_pretraining_driver_synthetic(tmp_path, dev):
    import run_pretraining_fcmf as drv
    hf = make_hf_dir(synth.TINY_CFG)
    out = str(tmp_path / "pt")
    drv.main(["--output_dir", out, "--pretrained_hf_model", hf, "--do_train", "--num_imgs", "2", "--num_rois", "5",
              "--train_batch_size", "3", "--synthetic_steps", "4", "--synthetic_dec_len", "6", "--num_train_epochs", "1",
              "--seed", "5"])
    ck = torch.load(os.path.join(out, "seed_5_iaog_model_last.pth"), map_location="cpu", weights_only=True)
    sd = ck["model_state_dict"]
    assert "decoder.blks.block0.attention1.w_kx" in sd and "decoder.pos_encoding.P" in sd
    # decoder.dense.weight stays tied to the encoder word embeddings; decoder.embedding is its own table
    assert torch.equal(sd["decoder.dense.weight"], sd["encoder.bert.cell.embeddings.word_embeddings.weight"])
    assert not torch.equal(sd["decoder.embedding.weight"], sd["decoder.dense.weight"])
    enc = {k: v for k, v in sd.items() if k.startswith("encoder.")}
    from fcmf_framework.fcmf_multimodal import FCMF
    m = FCMF(hf, num_imgs=2, num_roi=5)
    missing, unexpected = m.load_state_dict(enc, strict=False)      # IAOG -> fine-tune transfer
    assert not unexpected
