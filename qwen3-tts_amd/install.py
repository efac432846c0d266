"""Install the MI355X decode into a loaded Qwen3-TTS model.

The reference never calls the speech tokenizer itself: it calls `Qwen3TTSModel.generate_voice_clone` /
`generate_custom_voice` (reference server.py:245-251, models.py:86-101), which run the talker LM and then hand the
generated codes to the tokenizer's `decode`.  Replacing that one bound method is therefore the whole integration:

    model = Qwen3TTSModel.from_pretrained(path, device_map=..., dtype=...)     # reference models.py:43-48
    install_decoder(model)                                                     # <- the one added line
    wavs, sr = model.generate_voice_clone(...)                                 # unchanged

The `qwen-tts` package is not available where this was written, so nothing here relies on its attribute names: the
tokenizer object is found by shape (it owns a callable `decode` and a `decoder` sub-module whose state dict holds the
decoder's tensors), the architecture is read from that sub-module's config and cross-checked against tensor shapes,
and anything that does not line up raises instead of being guessed.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Iterable, Optional, Tuple

import torch

from .config import DecoderConfig, FRONTEND_MEAN_EMBEDDING, FRONTEND_SPLIT_RVQ
from .weights import param_specs

_ANCHOR = "decoder.0.conv.weight"            # first conv of the conv decoder: present for every front end


def _children(obj: Any) -> Iterable[Tuple[str, Any]]:
    if isinstance(obj, torch.nn.Module):
        yield from obj.named_children()
    d = getattr(obj, "__dict__", None)
    if isinstance(d, dict):
        for k, v in d.items():
            if k.startswith("__") or isinstance(v, (str, bytes, int, float, bool, torch.Tensor, type(None))):
                continue
            yield k, v


def find_speech_tokenizer(model: Any, max_depth: int = 4) -> Tuple[Any, Any]:
    """-> (tokenizer object, its decoder module).  Breadth-first over attributes; the tokenizer is the first object
    with a callable `decode` and a `decoder` attribute whose state dict contains the conv decoder's first weight."""
    seen, frontier = set(), [(model, 0)]
    while frontier:
        obj, depth = frontier.pop(0)
        if id(obj) in seen:
            continue
        seen.add(id(obj))
        dec = getattr(obj, "decoder", None)
        if callable(getattr(obj, "decode", None)) and dec is not None and callable(getattr(dec, "state_dict", None)):
            keys = dec.state_dict().keys()
            if any(k == _ANCHOR or k.endswith("." + _ANCHOR) for k in keys):
                return obj, dec
        if depth < max_depth:
            for _name, child in _children(obj):
                if not isinstance(child, (list, tuple, dict, set)):
                    frontier.append((child, depth + 1))
    raise LookupError("no speech tokenizer found: expected an object with a `decode` method and a `decoder` module whose "
                      f"state dict has '{_ANCHOR}'")


def decoder_weights(decoder_module: Any) -> Dict[str, torch.Tensor]:
    """The decoder's tensors under the names this package uses (the sibling Code2Wav's state-dict keys), with any
    common prefix of the module's own keys removed."""
    sd = dict(decoder_module.state_dict())
    prefix = None
    for k in sd:
        if k == _ANCHOR or k.endswith("." + _ANCHOR):
            prefix = k[: len(k) - len(_ANCHOR)]
            break
    if prefix is None:
        raise KeyError(f"state dict has no '{_ANCHOR}'")
    return {k[len(prefix):]: v.detach() for k, v in sd.items() if k.startswith(prefix)}


def _get(cfg_obj: Any, name: str, default: Any = None) -> Any:
    if isinstance(cfg_obj, dict):
        return cfg_obj.get(name, default)
    return getattr(cfg_obj, name, default)


# Sizes a tensor shape determines.  They are NEVER read from the model's config object: upstream configs reuse some of
# these names with another meaning (its `codebook_dim` is the RVQ in/out width, twice this package's code-book row width).
_SHAPE_FIELDS = ("frontend", "codebook_size", "codebook_dim", "rvq_out_dim", "num_quantizers", "num_semantic_quantizers",
                 "pre_conv_kernel", "latent_dim", "hidden_size", "io_proj", "num_hidden_layers", "intermediate_size", "decoder_dim")


def config_from_weights(cfg_obj: Any, w: Dict[str, torch.Tensor]) -> DecoderConfig:
    """Architecture of the decoder whose tensors are `w`.  Every size a tensor shape determines comes from the shape
    alone; the config object (or dict) only supplies what no shape shows (head count, window, rates, epsilons, rope
    theta), and what it says there is cross-checked against the shapes where one exists."""
    base = DecoderConfig()
    fields: Dict[str, Any] = {}
    for f in base.to_dict():
        if f in _SHAPE_FIELDS:
            continue
        v = _get(cfg_obj, f)
        if v is not None:
            fields[f] = tuple(v) if isinstance(v, (list, tuple)) else v

    def put(name: str, value: Any) -> None:
        if name in fields and fields[name] != value:
            raise ValueError(f"the weights imply {name} = {fields[name]!r} in one place and {value!r} in another")
        fields[name] = value

    if "code_embedding.weight" in w:
        put("frontend", FRONTEND_MEAN_EMBEDDING)
        rows, width = w["code_embedding.weight"].shape
        put("hidden_size", int(width))
        # one table of codebook_size * num_quantizers rows: the shape fixes only the product, the config splits it
        for f in ("codebook_size", "num_quantizers"):
            v = _get(cfg_obj, f)
            if v is not None:
                fields[f] = int(v)
        bins, nq = fields.get("codebook_size", base.codebook_size), fields.get("num_quantizers", base.num_quantizers)
        if bins * nq != int(rows):
            raise ValueError(f"code_embedding has {int(rows)} rows but codebook_size * num_quantizers = {bins} * {nq}")
        v = _get(cfg_obj, "num_semantic_quantizers")
        if v is not None:
            fields["num_semantic_quantizers"] = int(v)
    elif "quantizer.semantic.layers.0.embed_sum" in w:
        put("frontend", FRONTEND_SPLIT_RVQ)
        n_sem = sum(1 for k in w if k.startswith("quantizer.semantic.layers.") and k.endswith(".embed_sum"))
        n_ac = sum(1 for k in w if k.startswith("quantizer.acoustic.layers.") and k.endswith(".embed_sum"))
        bins, dim = w["quantizer.semantic.layers.0.embed_sum"].shape
        put("num_semantic_quantizers", n_sem)
        put("num_quantizers", n_sem + n_ac)
        put("codebook_size", int(bins))
        put("codebook_dim", int(dim))
        put("rvq_out_dim", int(w["quantizer.semantic.output_proj.weight"].shape[0]))
        lt, _cin, k = w["pre_conv.conv.weight"].shape
        put("latent_dim", int(lt))
        put("pre_conv_kernel", int(k))
    else:
        raise KeyError("state dict has neither 'code_embedding.weight' nor 'quantizer.semantic.layers.0.embed_sum'; "
                       "pass rename= to install_decoder if the checkpoint uses other key names")
    put("io_proj", "pre_transformer.input_proj.weight" in w)
    n_layers = 0
    while f"pre_transformer.layers.{n_layers}.input_layernorm.weight" in w:
        n_layers += 1
    put("num_hidden_layers", n_layers)
    if n_layers:
        put("hidden_size", int(w["pre_transformer.layers.0.input_layernorm.weight"].shape[0]))
        put("intermediate_size", int(w["pre_transformer.layers.0.mlp.gate_proj.weight"].shape[0]))
        qd = int(w["pre_transformer.layers.0.self_attn.q_proj.weight"].shape[0])
        kvd = int(w["pre_transformer.layers.0.self_attn.k_proj.weight"].shape[0])
        heads = int(fields.get("num_attention_heads", base.num_attention_heads))
        if qd % heads:
            raise ValueError(f"q_proj has {qd} rows, not a multiple of num_attention_heads = {heads}")
        hd = qd // heads
        if "head_dim" in fields and int(fields["head_dim"]) != hd:
            raise ValueError(f"decoder config says head_dim = {fields['head_dim']} but q_proj implies {hd}")
        fields["num_attention_heads"], fields["head_dim"] = heads, hd
        if kvd % hd:
            raise ValueError(f"k_proj has {kvd} rows, not a multiple of head_dim = {hd}")
        if "num_key_value_heads" in fields and int(fields["num_key_value_heads"]) != kvd // hd:
            raise ValueError(f"decoder config says num_key_value_heads = {fields['num_key_value_heads']} but k_proj implies {kvd // hd}")
        fields["num_key_value_heads"] = kvd // hd
    elif "hidden_size" not in fields:
        put("hidden_size", int(w[_ANCHOR].shape[1]))
    dd, lt, _k = w[_ANCHOR].shape
    put("decoder_dim", int(dd))
    put("latent_dim", int(lt))
    cfg = DecoderConfig.from_dict(fields)
    missing = [n for n, shape, _kind, _fan in param_specs(cfg) if n not in w or tuple(w[n].shape) != tuple(shape)]
    if missing:
        raise ValueError(f"{len(missing)} decoder tensors are missing or mis-shaped for the derived config, first: {missing[:4]}")
    return cfg


def rename_weights(w: Dict[str, torch.Tensor], rename: Optional[Dict[str, str]]) -> Dict[str, torch.Tensor]:
    """Apply a key-rename map: each (old, new) pair replaces the substring `old` by `new` in every key that contains
    it.  Unknown key names are never guessed; a checkpoint that names its tensors differently is loaded by passing the
    map explicitly."""
    if not rename:
        return w
    out: Dict[str, torch.Tensor] = {}
    for k, v in w.items():
        for old, new in rename.items():
            if old in k:
                k = k.replace(old, new)
        if k in out:
            raise ValueError(f"rename map sends two tensors to '{k}'")
        out[k] = v
    return out


def install_decoder(model: Any, precision: str = "bf16", device: Optional[Any] = None,
                    decoder_factory: Optional[Callable[..., Any]] = None, rename: Optional[Dict[str, str]] = None,
                    **decoder_kwargs: Any) -> Any:
    """Find the model's speech tokenizer, build a `Decoder` from its weights on `device` (default: where those weights
    live) and replace the tokenizer's `decode` with it.  The previous bound method stays reachable as
    `tokenizer.reference_decode`.  `rename` maps key substrings of the checkpoint to the names this package uses
    (see rename_weights).  Returns the new decoder."""
    tok, dec_mod = find_speech_tokenizer(model)
    w = rename_weights(decoder_weights(dec_mod), rename)
    cfg = config_from_weights(getattr(dec_mod, "config", None) or getattr(tok, "decoder_config", None) or {}, w)
    if device is None:
        device = next(iter(w.values())).device
        if torch.device(device).type != "cuda":
            raise RuntimeError("the model's decoder weights are not on a GPU; pass device='cuda:N' (there is no CPU fallback)")
    if decoder_factory is None:
        from .decoder import Decoder as decoder_factory          # noqa: N813  (needs the built library)
    names = {n for n, *_ in param_specs(cfg)}
    names |= {k for k in w if k.startswith("quantizer.") and k.endswith(".input_proj.weight")}      # encode direction, optional
    new = decoder_factory(cfg, {k: v for k, v in w.items() if k in names}, device=device, precision=precision, **decoder_kwargs)
    if not hasattr(tok, "reference_decode"):
        tok.reference_decode = tok.decode
    tok.decode = new.decode
    tok.q3c_decoder = new
    return new


def uninstall_decoder(model: Any) -> None:
    tok, _ = find_speech_tokenizer(model)
    if hasattr(tok, "reference_decode"):
        tok.decode = tok.reference_decode
        del tok.reference_decode
        tok.q3c_decoder = None


class ModelCodeSource:
    """A `server.CodeSource` made of a loaded Qwen3-TTS model: text -> the codes its talker LM generates.

    The reference's calls return finished audio (`generate_voice_clone(...) -> (wavs, sr)`, reference server.py:245-251); the
    one place the codes are visible from outside the `qwen-tts` package is where the model hands them to the tokenizer's
    `decode`.  While a call of this object is in flight that `decode` is a recorder: it keeps the codes and returns silence
    of the length the real decode would produce, so the model's own generation runs to its end, nothing is decoded twice,
    and the server's streaming / batched decode gets the codes.  One generation at a time per model (the swap is a lock's
    critical section), as the reference serves them.

        src = ModelCodeSource(model, decoder.cfg)                       # after install_decoder(model)
        engine = server.Engine(decoder, src)
    """

    def __init__(self, model: Any, cfg: DecoderConfig, method: str = "generate_voice_clone", prompt_cache: int = 16):
        import threading
        self.model, self.cfg, self.method = model, cfg, method
        self._tok, _ = find_speech_tokenizer(model)
        self._lock = threading.Lock()
        self._prompts: Dict[tuple, Any] = {}         # (ref_audio, ref_text, x_vector_only_mode) -> voice clone prompt
        self._prompt_cap = int(prompt_cache)

    def _with_prompt(self, kw: Dict[str, Any]) -> Dict[str, Any]:
        """The reference computes the voice-clone prompt (speaker embedding, reference codes) ONCE per request and passes it to
        every sentence's generation (`create_voice_clone_prompt`, then `voice_clone_prompt=` with `non_streaming_mode=True`,
        server.py:228-251, 298-372), falling back to the direct `ref_audio` call when that fails.  Here the prompt is kept
        per (reference audio, transcript, mode) for the most recent few, so the sentences of a request - and later requests
        with the same voice - share it.  Models without `create_voice_clone_prompt`, calls without a `ref_audio` string, and
        any failure take the direct call."""
        make = getattr(self.model, "create_voice_clone_prompt", None)
        ref = kw.get("ref_audio")
        if self.method != "generate_voice_clone" or make is None or not isinstance(ref, str) or "voice_clone_prompt" in kw:
            return kw
        key = (ref, kw.get("ref_text"), bool(kw.get("x_vector_only_mode", True)))
        prompt = self._prompts.pop(key, None)
        if prompt is None:
            try:
                prompt = make(ref_audio=ref, ref_text=kw.get("ref_text"), x_vector_only_mode=key[2])
                item = prompt[0] if prompt else None
                if item is None or getattr(item, "ref_spk_embedding", None) is None:
                    raise ValueError("voice clone prompt is empty or has no speaker embedding")
            except Exception as e:                              # as the reference: "falling back to direct ref_audio mode"
                print(f"[ModelCodeSource] voice clone prompt failed ({e}); falling back to the direct ref_audio call")
                return kw
            while len(self._prompts) >= self._prompt_cap:
                self._prompts.pop(next(iter(self._prompts)))
        self._prompts[key] = prompt                             # (re)inserted last: most recently used
        out = {k: v for k, v in kw.items() if k not in ("ref_audio", "ref_text", "x_vector_only_mode")}
        out.update(voice_clone_prompt=prompt, non_streaming_mode=True)
        return out

    def _items(self, encoded: Any) -> list:
        """The wrapper's input forms (a [T, Q] / [Q, T] matrix, a [B, ...] tensor, {"audio_codes": ...}, or a list of those)
        -> list of [Q, T] int64 tensors on the host."""
        Q = self.cfg.num_quantizers
        if isinstance(encoded, dict):
            encoded = encoded["audio_codes"]
        if isinstance(encoded, (list, tuple)):
            return [c for e in encoded for c in self._items(e)]
        t = torch.as_tensor(encoded).detach().to("cpu", torch.int64)
        if t.dim() == 3:
            return [c for e in t for c in self._items(e)]
        if t.dim() != 2 or Q not in t.shape:
            raise ValueError(f"Expected codes with a dimension of {Q} code books, got {tuple(t.shape)}")
        if t.shape[1] == Q:                                    # [T, Q], the item form of the wrapper (a square matrix reads as this)
            return [t.t().contiguous()]
        return [t]

    def __call__(self, text: str, seed: Optional[int] = None, **kw: Any) -> torch.Tensor:
        import numpy as np
        seen: list = []

        def recorder(encoded: Any, *a: Any, **k: Any):
            items = self._items(encoded)
            seen.extend(items)
            return [np.zeros(self.cfg.chunked_output_length(int(c.shape[1])), dtype=np.float32) for c in items], self.cfg.sample_rate

        with self._lock:
            real = self._tok.decode
            self._tok.decode = recorder
            try:
                if seed is not None:                        # as the reference seeds a request (server.py:348-351)
                    torch.manual_seed(int(seed))
                    if torch.cuda.is_available():
                        torch.cuda.manual_seed(int(seed))
                getattr(self.model, self.method)(text=text, **self._with_prompt(kw))
            finally:
                self._tok.decode = real
        if not seen:
            raise RuntimeError(f"{self.method} finished without handing any codes to the speech tokenizer's decode")
        return seen[0] if len(seen) == 1 else torch.cat(seen, dim=1)
