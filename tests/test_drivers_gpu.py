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
    # ---- the two ResNet-152 extractors are checkpointed beside the model under the reference's names (:557-563) ----------
    for tag in ("resimg", "resroi"):
        rk = torch.load(os.path.join(out, f"seed_4_{tag}_model_last.pth"), map_location="cpu", weights_only=True)
        assert set(rk) >= {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict"}
        sd = rk["model_state_dict"]
        assert "resnet.conv1.weight" in sd and "resnet.layer3.35.bn3.running_var" in sd and "resnet.bn1.num_batches_tracked" in sd
        # train()-mode BatchNorm ran inside the step (reference :431): the running statistics have left their initial values
        assert int(sd["resnet.bn1.num_batches_tracked"]) > 0 and not torch.equal(sd["resnet.bn1.running_mean"], torch.zeros(64))
    # ---- test-set evaluation after training (reference :567-694): the two result files, in the reference's format ---------
    res = open(os.path.join(out, "test_results_fcmf.txt")).read().splitlines()
    assert res[0] == "***** Test results *****" and res[-1].startswith("Average F1: ")
    aspects = ['Location', 'Food', 'Room', 'Facilities', 'Service', 'Public_area']
    import re
    for a, line in zip(aspects, res[1:7]):
        assert re.fullmatch(rf"{a} - P: \d\.\d{{4}}, R: \d\.\d{{4}}, F1: \d\.\d{{4}}", line), line
    fm = open(os.path.join(out, "test_predictions_formatted.txt"), encoding="utf-8").read()
    assert fm.startswith("TEST DETAILED PREDICTIONS\nAverage Macro F1: ") and "Sentence 0: synthetic review" in fm
    assert fm.count("{\n") == fm.count("}\n") == 2 and fm.count("   predict: ") == 2 * 6       # 1 test batch x 2 reviews x 6 aspects
    assert all(v in ("None", "Negative", "Neutral", "Positive") for v in re.findall(r"predict: (\w+)", fm))
    # ---- resume (round-2 advisor finding): the extractors come back from their own checkpoints, BatchNorm buffers included --
    import run_multimodal_fcmf as drv2
    seen = {}
    orig = drv2.load_resnets

    def spy(path, rimg, rroi, *a, **kw):
        r = orig(path, rimg, rroi, *a, **kw)
        seen["paths"] = r
        seen["img_bn"] = rimg.state_dict()["resnet.bn1.running_mean"].detach().cpu().clone()
        return r
    drv2.load_resnets = spy
    try:
        drv2.main(["--output_dir", out, "--pretrained_hf_model", hf, "--do_train", "--num_imgs", "2", "--num_rois", "2",
                   "--train_batch_size", "2", "--gradient_accumulation_steps", "1", "--synthetic_steps", "2",
                   "--synthetic_pixels", "64", "--max_seq_length", "16", "--seed", "4", "--num_train_epochs", "1", "--bf16",
                   "--resume_from_checkpoint", os.path.join(out, "seed_4_fcmf_model_last.pth")])
    finally:
        drv2.load_resnets = orig
        ops.set_compute_dtype(torch.float32)
    assert len(seen["paths"]) == 2 and seen["paths"][0].endswith("seed_4_resimg_model_last.pth")
    want = torch.load(os.path.join(out, "seed_4_resimg_model_last.pth"), map_location="cpu", weights_only=True)
    assert torch.equal(seen["img_bn"], want["model_state_dict"]["resnet.bn1.running_mean"])


def test_pretraining_driver_fine_tune_cnn_moves_the_trunks(tmp_path, dev):
    """--fine_tune_cnn (reference run_pretraining_fcmf.py:203-207): the parameters of both ResNet-152 extractors are in the
    AdamW groups, so a step CHANGES the convolution weights; without the flag the same seed leaves them at their initial
    values while the train()-mode BatchNorm statistics still drift.  Both runs write the extractors' checkpoints (:457-459);
    a resume restores them through the `iaog_model` -> `resimg_model` rewrite (:244-255)."""
    import run_pretraining_fcmf as drv
    from fcmf_framework import ops
    hf = make_hf_dir(synth.TINY_CFG)
    common = ["--pretrained_hf_model", hf, "--do_train", "--num_imgs", "2", "--num_rois", "2", "--train_batch_size", "2",
              "--synthetic_steps", "2", "--synthetic_dec_len", "5", "--synthetic_pixels", "64", "--num_train_epochs", "1",
              "--seed", "6", "--bf16", "--learning_rate", "1e-3"]
    try:
        outs = {}
        for name, extra in (("frozen", []), ("tuned", ["--fine_tune_cnn"])):
            out = str(tmp_path / name)
            drv.main(["--output_dir", out] + common + extra)
            outs[name] = {t: torch.load(os.path.join(out, f"seed_6_{t}_model_last.pth"), map_location="cpu", weights_only=True)
                          for t in ("resimg", "resroi", "iaog")}
        for t in ("resimg", "resroi"):
            f, g = outs["frozen"][t]["model_state_dict"], outs["tuned"][t]["model_state_dict"]
            moved = [k for k in f if k.endswith("conv1.weight") or k.endswith("conv2.weight") or k.endswith("bn2.weight")]
            assert len(moved) > 100
            n_changed = sum(not torch.equal(f[k], g[k]) for k in moved)
            assert n_changed == len(moved), (t, n_changed, len(moved))                   # every trunk weight received an update
            assert all(torch.isfinite(g[k]).all() for k in g if g[k].is_floating_point())
            assert int(f["resnet.bn1.num_batches_tracked"]) > 0                          # statistics drift either way
        # the optimizer state of the tuned run covers the trunk parameters too
        n_model = len(outs["frozen"]["iaog"]["optimizer_state_dict"]["state"])
        assert len(outs["tuned"]["iaog"]["optimizer_state_dict"]["state"]) > n_model + 800      # 2 x ~465 trunk tensors (the unused fc layers receive no gradient)
        # resume of the tuned run
        out = str(tmp_path / "tuned")
        drv.main(["--output_dir", out] + common + ["--fine_tune_cnn", "--num_train_epochs", "2", "--resume_from_checkpoint",
                                                   os.path.join(out, "seed_6_iaog_model_last.pth")])
        ck2 = torch.load(os.path.join(out, "seed_6_iaog_model_last.pth"), map_location="cpu", weights_only=True)
        assert ck2["epoch"] == 1
    finally:
        ops.set_compute_dtype(torch.float32)


def test_device_prefetcher_hands_batches_to_hbm(tmp_path, dev):
    """SURVEY section 8f.3, the device hand-off of the batch producer: the reference's tuple layout with float64 ROI crops
    and boxes (vimacsa_dataset.py:188-202) and the bf16 feature cache arrive in HBM through pinned memory on a copy
    stream -- values exact, order kept, float64 pixels -> float32 on the device (the reference's `.float()`), float64
    boxes left alone, bf16 features bit-identical, strings passed through."""
    from device_prefetch import DevicePrefetcher
    from feature_cache import FeatureCache, FeatureCacheWriter
    g = torch.Generator().manual_seed(0)
    host = []
    for i in range(5):
        host.append((torch.randn(2, 2, 3, 16, 16, generator=g),                                   # images, float32
                     torch.randn(2, 2, 3, 3, 16, 16, generator=g, dtype=torch.float64),           # ROI crops, float64
                     torch.rand(2, 2, 3, 4, generator=g, dtype=torch.float64),                    # boxes, float64
                     torch.randint(0, 100, (2, 6, 16), generator=g), torch.full((2, 6), i), [f"text {i}a", f"text {i}b"]))
    pf = DevicePrefetcher(iter(host), dev)
    assert pf.copy_stream != torch.cuda.current_stream(dev)
    n = 0
    for i, (img, roi, box, ids, lab, texts) in enumerate(pf):
        assert img.is_cuda and roi.dtype == torch.float32 and box.dtype == torch.float64 and ids.dtype == torch.int64
        assert torch.equal(img.cpu(), host[i][0]) and torch.equal(roi.cpu(), host[i][1].float())
        assert torch.equal(box.cpu(), host[i][2]) and torch.equal(ids.cpu(), host[i][3]) and int(lab[0, 0]) == i
        assert texts == host[i][5]
        y = (img * 2).sum()                     # consume on the compute stream while the next copy is in flight
        n += 1
    assert n == 5 and pf.bytes_copied == sum(t.numel() * t.element_size() for b in host for t in b[:5])
    # ---- --num_imgs 3: the box tensor is [B, 3, num_rois, 4] -- it must stay float64 (round-3 advisor finding), by shape and by field
    b3 = (torch.randn(2, 3, 3, 8, 8), torch.randn(2, 3, 5, 3, 8, 8, dtype=torch.float64), torch.rand(2, 3, 5, 4, dtype=torch.float64))
    for kw in ({}, {"float32_fields": (1,)}):
        (img, roi, box), = list(DevicePrefetcher(iter([b3]), dev, **kw))
        assert roi.dtype == torch.float32 and box.dtype == torch.float64 and torch.equal(box.cpu(), b3[2])
    (d,), = [tuple([x]) for x in DevicePrefetcher(iter([{"roi": b3[1], "box": b3[2]}]), dev, float32_fields=("roi",))]
    assert d["roi"].dtype == torch.float32 and d["box"].dtype == torch.float64
    # ---- bf16 feature cache -> HBM ------------------------------------------------------------------------------
    path = str(tmp_path / "cache")
    w = FeatureCacheWriter(path, 4, 2, 3)
    vis = torch.randn(4, 2, 49, 2048, generator=g); roi = torch.randn(4, 2, 3, 2048, generator=g); co = torch.rand(4, 2, 3, 4, generator=g)
    w.append(vis[:2], roi[:2], co[:2]); w.append(vis[2:], roi[2:], co[2:]); w.close()
    cache = FeatureCache(path)

    def batches():
        for s in (0, 2):
            items = [cache[s], cache[s + 1]]
            yield tuple(torch.stack([it[k] for it in items]) for k in range(3))
    got = list(DevicePrefetcher(batches(), dev))
    assert len(got) == 2 and got[0][0].dtype == torch.bfloat16 and got[0][0].is_cuda
    assert torch.equal(torch.cat([b[0] for b in got]).cpu(), vis.bfloat16())
    assert torch.equal(torch.cat([b[1] for b in got]).cpu(), roi.bfloat16())
    assert torch.equal(torch.cat([b[2] for b in got]).cpu(), co)
    # an exception in the producer surfaces in the consumer

    def bad():
        yield (torch.zeros(2),)
        raise ValueError("producer failed")
    with pytest.raises(ValueError, match="producer failed"):
        list(DevicePrefetcher(bad(), dev))


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


def test_bare_bench_gpus2_runs_the_real_step_on_two_ranks(dev):
    """`python bench.py --gpus 2` with no launcher (what the scaling driver types): bench.py starts the two ranks itself; here both
    share the box's single MI355X (FCMF_BENCH_SINGLE_DEVICE=1) over gloo, so the timings mean nothing -- what is checked is the
    path: rc 0, ONE line with n_gpus = 2, the gradient exchange in launch groups (several per step, i.e. overlapped with backward),
    and the batched weight-gradient kernel still multiplying several layers per launch under data parallelism (round 3 flushed
    the queue per bucket: one matrix per launch).  reference: run_multimodal_fcmf.py:126-131,237-240"""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FCMF_BENCH_SINGLE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--batch", "8", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["dp_ranks_seen"] == 2 and doc["dp_backend"] == "gloo" and doc["config"]["global_batch"] == 16
    dp = doc["dp"]
    assert dp["world"] == 2 and dp["buckets"] >= 10 and dp["bytes_per_step"] > 6e8
    assert 2 <= dp["launch_groups"] < dp["buckets"], dp
    assert dp["dw_matrices_per_batched_launch"] >= 2.5, dp
    assert doc["value"] > 0 and doc["loss"] == doc["loss"]


def test_bench_over_rccl_prints_exactly_one_json_line(dev):
    """the data-parallel bench path over RCCL as far as one GPU allows (--dp-one-rank: process group "nccl" of one rank bound to
    the device, arena + hooks + launch groups + one RCCL all-reduce per bucket on the side stream + finish()).  RCCL writes a
    version banner to STDOUT when the first communicator is created; the bench contract is ONE JSON line there (what the scaling
    driver parses), so stdout must hold that line and nothing else.  reference: run_multimodal_fcmf.py:126-131,237-240"""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dp-one-rank", "--steps", "2", "--warmup", "1", "--batch", "8",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1 and out[0].startswith("{"), r.stdout[:2000]
    doc = json.loads(out[0])
    assert doc["n_gpus"] == 1 and doc["dp_backend"] == "nccl" and doc["dp_ranks_seen"] == 1
    dp = doc["dp"]
    assert dp["buckets"] >= 10 and 2 <= dp["launch_groups"] < dp["buckets"] and dp["exposed_comm_ms_per_step"] is not None
    assert doc["value"] > 0 and doc["loss"] == doc["loss"]
