"""Checkpoint loaders (reference seal/utils.py:42-50, 31-39): fairseq / lightning
state dicts -> the HF ``BartForConditionalGeneration`` the decoder reads."""
import torch


_DERIVED_KEYS = {"lm_head.weight", "final_logits_bias", "model.encoder.embed_tokens.weight", "model.decoder.embed_tokens.weight",
                 "model.shared.weight", "model.decoder.output_projection.weight", "model.encoder.embed_positions._float_tensor",
                 "model.decoder.embed_positions._float_tensor"}


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
    # the reference loads strictly (seal/utils.py:39,50); here strict up to the tensors that are functions of the
    # shared embedding or absent from fairseq checkpoints -- anything else missing/unexpected is a wrong checkpoint
    res = model.load_state_dict(state_dict, strict=False)
    bad = [k for k in res.missing_keys if k not in _DERIVED_KEYS] + [k for k in res.unexpected_keys if k not in _DERIVED_KEYS]
    if bad:
        raise RuntimeError(f"checkpoint does not match the model: missing/unexpected keys {sorted(bad)[:8]}"
                           f"{' ...' if len(bad) > 8 else ''}")
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
    """lightning wrapper ({"state_dict": {"model.<hf key>": ...}}) or, as the reference's loader expects
    (seal/utils.py:31-39), a plain HF state dict"""
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    have = set(model.state_dict())
    out = {(k[len("model."):] if k not in have and k.startswith("model.") and k[len("model."):] in have else k): v
           for k, v in sd.items()}
    _tie_and_load(model, out)
