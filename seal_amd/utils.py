"""Checkpoint loaders (reference seal/utils.py:42-50, 31-39): fairseq / lightning
state dicts -> the HF ``BartForConditionalGeneration`` the decoder reads."""
import torch


def _tie_and_load(model, state_dict):
    # fairseq checkpoints carry the shared embedding once and no lm_head; HF has one
    # extra (mask) row after resize_token_embeddings -> append a zero row
    emb = state_dict["model.shared.weight"] if "model.shared.weight" in state_dict else None
    if emb is not None:
        rows = model.get_input_embeddings().weight.shape[0]
        if emb.shape[0] < rows:
            emb = torch.cat([emb, torch.zeros(rows - emb.shape[0], emb.shape[1], dtype=emb.dtype)], 0)
        for k in ("model.shared.weight", "model.encoder.embed_tokens.weight", "model.decoder.embed_tokens.weight", "lm_head.weight"):
            state_dict[k] = emb
    model.load_state_dict(state_dict, strict=False)
    model.tie_weights()


def load_state_dict_from_fairseq_checkpoint(model, path):
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt["model"] if "model" in ckpt else ckpt
    out = {}
    for k, v in sd.items():
        if k.endswith("version") or k.endswith("_float_tensor"):
            continue
        k = k.replace("encoder.embed_tokens", "shared") if k == "encoder.embed_tokens.weight" else k
        out["model." + k if not k.startswith("model.") else k] = v
    _tie_and_load(model, out)


def load_state_dict_from_lightning_checkpoint(model, path):
    sd = torch.load(path, map_location="cpu")["state_dict"]
    out = {(k[len("model."):] if k.startswith("model.model.") or k.startswith("model.lm_head") else k): v for k, v in sd.items()}
    _tie_and_load(model, out)
