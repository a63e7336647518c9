"""shared test helpers (tests may use oracle/, the product may not)"""
import tempfile

import torch

import synthetic_data as synth


def make_hf_dir(cfg):
    from fcmf_framework.roberta import RobertaConfig, RobertaModel
    d = tempfile.mkdtemp(prefix="hf_")
    RobertaModel(RobertaConfig(**cfg)).save_pretrained(d)
    return d


def build_fcmf(cfg, NI, NR, device, num_labels=4, seed=0):
    """product FCMF with the deterministic synthetic weights of oracle/synth.py"""
    from fcmf_framework.fcmf_multimodal import FCMF
    model = FCMF(make_hf_dir(cfg), num_labels=num_labels, num_imgs=NI, num_roi=NR)
    P = synth.synth_params(synth.fcmf_param_shapes(cfg, num_labels), seed)
    missing, unexpected = model.load_state_dict(P, strict=True)
    return model.to(device), P


def batch_to(batch, device):
    return {k: v.to(device) for k, v in batch.items()}


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def max_err(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()
