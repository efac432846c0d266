"""Generate the golden fixtures under tests/golden/ from the in-container sibling implementation.

Run from the repository root in the BUILD container (needs `transformers` 5.15.0):

    python tests/golden/make_goldens.py

The reference repository holds no golden vector for the decode and its own implementation of it
(the `qwen-tts` PyPI package) is absent, so the expected values here come from library code that is
already in the container and implements the same family of arithmetic:
  transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe  (Code2Wav and its parts)
  transformers.models.mimi.modeling_mimi                       (split residual VQ)
Every expected array in a fixture is the SIBLING's output, never this repository's oracle's; the
script also asserts that the oracle agrees before it writes anything.  Inputs and weights are drawn
from seeded generators; fixtures store the inputs, the expected outputs and a float64 checksum of
the weights they were made with.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import qwen3_tts_amd as q3  # noqa: E402
import decoder_oracle as oracle  # noqa: E402

from transformers.models.mimi.configuration_mimi import MimiConfig  # noqa: E402
from transformers.models.mimi import modeling_mimi as mimi  # noqa: E402
from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import Qwen3OmniMoeCode2WavConfig  # noqa: E402
from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as sib  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def wsum(weights) -> float:
    return float(sum(t.double().abs().sum().item() for t in weights.values()))


def save(name, meta, **arrays):
    arrays = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    path = os.path.join(OUT, name + ".npz")
    np.savez(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print(f"wrote {path}  {os.path.getsize(path) / 1024:.0f} KiB")


# ---------------------------------------------------------------------------------------------
def make_rvq():
    for tag, kw in (("rvq_tiny", dict(codebook_size=32, num_quantizers=4, num_semantic_quantizers=1, codebook_dim=8, rvq_out_dim=16)),
                    ("rvq_q16", dict(codebook_size=256, num_quantizers=16, num_semantic_quantizers=1, codebook_dim=16, rvq_out_dim=32)),
                    ("rvq_sem2", dict(codebook_size=64, num_quantizers=8, num_semantic_quantizers=2, codebook_dim=8, rvq_out_dim=24))):
        cfg = q3.tiny_config(**kw)
        w = q3.synthetic_weights(cfg, seed=11)
        mc = MimiConfig(codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim,
                        vector_quantization_hidden_dimension=cfg.codebook_dim, hidden_size=cfg.rvq_out_dim,
                        num_quantizers=cfg.num_quantizers, num_semantic_quantizers=cfg.num_semantic_quantizers)
        m = mimi.MimiSplitResidualVectorQuantizer(mc).eval()
        sd = {}
        for g, attr in (("semantic", "semantic_residual_vector_quantizer"), ("acoustic", "acoustic_residual_vector_quantizer")):
            n = cfg.num_semantic_quantizers if g == "semantic" else cfg.num_quantizers - cfg.num_semantic_quantizers
            for i in range(n):
                sd[f"{attr}.layers.{i}.codebook.embed_sum"] = w[f"quantizer.{g}.layers.{i}.embed_sum"]
                sd[f"{attr}.layers.{i}.codebook.cluster_usage"] = w[f"quantizer.{g}.layers.{i}.cluster_usage"]
                sd[f"{attr}.layers.{i}.codebook.initialized"] = torch.tensor([1.0])
            sd[f"{attr}.output_proj.weight"] = w[f"quantizer.{g}.output_proj.weight"]
            sd[f"{attr}.input_proj.weight"] = torch.zeros_like(w[f"quantizer.{g}.output_proj.weight"]).transpose(0, 1).contiguous()
        m.load_state_dict(sd, strict=True)
        codes = q3.random_codes(cfg, 3, 37, seed=5)
        with torch.no_grad():
            want = m.decode(codes)                                   # [B, out, T]
            sums = []
            for attr, lo, hi in (("semantic_residual_vector_quantizer", 0, cfg.num_semantic_quantizers),
                                 ("acoustic_residual_vector_quantizer", cfg.num_semantic_quantizers, cfg.num_quantizers)):
                rq = getattr(m, attr)
                acc = torch.full((), 0.0)
                for i, ind in enumerate(codes[:, lo:hi].transpose(0, 1)):
                    acc = acc + rq.layers[i].decode(ind)              # [B, D, T]
                sums.append(acc.permute(0, 2, 1))
            sums = torch.stack(sums, 1)                               # [B, G, T, D]
        got, gsums = oracle.rvq_dequant(cfg, w, codes)
        assert torch.equal(torch.stack(gsums, 1), sums), "oracle group sums are not bit-exact vs mimi"
        assert torch.allclose(got.permute(0, 2, 1), want, atol=1e-5, rtol=1e-5)
        nclamped = sum(int((w[k] < cfg.rvq_eps).sum()) for k in w if k.endswith("cluster_usage"))
        assert nclamped > 0
        save(tag, dict(config=cfg.to_dict(), weight_seed=11, weight_abs_sum=wsum(w), clamped_rows=nclamped),
             codes=codes, group_sums=sums, out=want.permute(0, 2, 1).contiguous())


def make_rvq_encode():
    """The quantiser's ENCODE direction: mimi's split RVQ `encode` on seeded embeddings (its input_proj random, its code
    books the synthetic ones): expected codes, and the oracle must reproduce them exactly."""
    g = torch.Generator().manual_seed(31)
    for tag, kw, B, T in (("rvq_encode_tiny", dict(codebook_size=32, num_quantizers=4, num_semantic_quantizers=1, codebook_dim=8, rvq_out_dim=16), 2, 41),
                          ("rvq_encode_q16", dict(codebook_size=256, num_quantizers=16, num_semantic_quantizers=1, codebook_dim=16, rvq_out_dim=32), 2, 29)):
        cfg = q3.tiny_config(**kw)
        w = q3.synthetic_weights(cfg, seed=13)
        mc = MimiConfig(codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim,
                        vector_quantization_hidden_dimension=cfg.codebook_dim, hidden_size=cfg.rvq_out_dim,
                        num_quantizers=cfg.num_quantizers, num_semantic_quantizers=cfg.num_semantic_quantizers)
        m = mimi.MimiSplitResidualVectorQuantizer(mc).eval()
        sd, extra = {}, {}
        for grp, attr in (("semantic", "semantic_residual_vector_quantizer"), ("acoustic", "acoustic_residual_vector_quantizer")):
            n = cfg.num_semantic_quantizers if grp == "semantic" else cfg.num_quantizers - cfg.num_semantic_quantizers
            for i in range(n):
                sd[f"{attr}.layers.{i}.codebook.embed_sum"] = w[f"quantizer.{grp}.layers.{i}.embed_sum"]
                sd[f"{attr}.layers.{i}.codebook.cluster_usage"] = w[f"quantizer.{grp}.layers.{i}.cluster_usage"]
                sd[f"{attr}.layers.{i}.codebook.initialized"] = torch.tensor([1.0])
            sd[f"{attr}.output_proj.weight"] = w[f"quantizer.{grp}.output_proj.weight"]
            ip = torch.randn(cfg.codebook_dim, cfg.rvq_out_dim, 1, generator=g) * (cfg.rvq_out_dim ** -0.5) * 3.0
            sd[f"{attr}.input_proj.weight"] = ip
            extra[f"quantizer.{grp}.input_proj.weight"] = ip
        m.load_state_dict(sd, strict=True)
        emb = torch.randn(B, cfg.rvq_out_dim, T, generator=g) * 2.0
        with torch.no_grad():
            want = m.encode(emb).permute(1, 0, 2).contiguous()            # [K, B, T] -> [B, K, T]
        got = oracle.rvq_encode(cfg, {**w, **extra}, emb)
        assert torch.equal(got, want), "oracle rvq_encode differs from mimi"
        assert want.shape == (B, cfg.num_quantizers, T) and int(want.max()) < cfg.codebook_size
        assert len(torch.unique(want)) > cfg.codebook_size // 4                # the code books are really used
        save(tag, dict(config=cfg.to_dict(), weight_seed=13, weight_abs_sum=wsum(w)),
             embeddings=emb, codes=want, **{k.replace(".", "__"): v for k, v in extra.items()})


# ---------------------------------------------------------------------------------------------
def mimi_for_encoder(ecfg, cfg, w):
    """The sibling's MimiModel with the encoder sizes of `ecfg`, the quantiser sizes of `cfg` and the tensors of `w`."""
    import math
    hop = math.prod(ecfg.ratios)
    mc = MimiConfig(audio_channels=1, hidden_size=ecfg.hidden_size, num_filters=ecfg.num_filters, num_residual_layers=1,
                    upsampling_ratios=list(ecfg.ratios[::-1]), kernel_size=ecfg.kernel_size, last_kernel_size=ecfg.last_kernel_size,
                    residual_kernel_size=ecfg.residual_kernel_size, dilation_growth_rate=2, use_causal_conv=True, pad_mode="constant",
                    compress=ecfg.compress, trim_right_ratio=1.0, codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim,
                    vector_quantization_hidden_dimension=cfg.codebook_dim, num_quantizers=cfg.num_quantizers,
                    num_semantic_quantizers=cfg.num_semantic_quantizers, sampling_rate=24000, frame_rate=24000 / hop / ecfg.downsample_stride,
                    num_hidden_layers=ecfg.num_hidden_layers, num_attention_heads=ecfg.num_attention_heads,
                    num_key_value_heads=ecfg.num_attention_heads, head_dim=ecfg.head_dim, intermediate_size=ecfg.intermediate_size,
                    hidden_act="gelu", norm_eps=ecfg.norm_eps, sliding_window=ecfg.sliding_window, rope_theta=ecfg.rope_theta,
                    use_conv_shortcut=False, use_cache=False, use_streaming=False, attn_implementation="eager",
                    upsample_groups=ecfg.hidden_size)
    m = mimi.MimiModel(mc).eval()
    sd = {k: v for k, v in w.items() if k.startswith(("encoder.", "encoder_transformer.", "downsample."))}
    for grp, attr in (("semantic", "semantic_residual_vector_quantizer"), ("acoustic", "acoustic_residual_vector_quantizer")):
        n = cfg.num_semantic_quantizers if grp == "semantic" else cfg.num_quantizers - cfg.num_semantic_quantizers
        for i in range(n):
            sd[f"quantizer.{attr}.layers.{i}.codebook.embed_sum"] = w[f"quantizer.{grp}.layers.{i}.embed_sum"]
            sd[f"quantizer.{attr}.layers.{i}.codebook.cluster_usage"] = w[f"quantizer.{grp}.layers.{i}.cluster_usage"]
            sd[f"quantizer.{attr}.layers.{i}.codebook.initialized"] = torch.tensor([1.0])
        sd[f"quantizer.{attr}.input_proj.weight"] = w[f"quantizer.{grp}.input_proj.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not [k for k in missing if k.startswith(("encoder.", "encoder_transformer.", "downsample.")) or "input_proj" in k or "codebook" in k], missing
    return m


def make_encoder():
    """The tokenizer's ENCODE side: the sibling MimiModel.encode (conv encoder, transformer, stride-2 conv, split-RVQ encode) on
    seeded audio, at small sizes; expected = its codes and the embeddings it quantises.  Lengths include ones that no stride
    divides (every strided conv pads on the right) and one shorter than a frame."""
    import importlib
    E = importlib.import_module("qwen3-tts_amd.encoder")
    import encoder_oracle as eo
    g = torch.Generator().manual_seed(57)
    for tag, ekw, dkw, lengths in (
            ("encoder_tiny", dict(), dict(codebook_size=32, num_quantizers=4, num_semantic_quantizers=1, codebook_dim=8, rvq_out_dim=32), (240, 101, 7)),
            ("encoder_wide", dict(num_filters=16, ratios=(2, 2, 3), hidden_size=64, num_attention_heads=2, head_dim=32, intermediate_size=128,
                                  num_hidden_layers=2, sliding_window=9),
             dict(codebook_size=64, num_quantizers=8, num_semantic_quantizers=1, codebook_dim=16, rvq_out_dim=64), (1200, 613))):
        ecfg = E.tiny_encoder_config(**ekw)
        cfg = q3.tiny_config(**dkw)
        w = {**q3.synthetic_weights(cfg, seed=17), **E.synthetic_encoder_weights(ecfg, cfg.codebook_dim, seed=19)}
        m = mimi_for_encoder(ecfg, cfg, w)
        arrays = {}
        for L in lengths:
            audio = torch.randn(2, 1, L, generator=g) * 0.5
            with torch.no_grad():
                emb = m.encoder(audio)
                emb = m.encoder_transformer(emb.transpose(1, 2), return_dict=False)[0].transpose(1, 2)
                emb = m.downsample(emb)
                codes = m.encode(audio, return_dict=False)[0]
            assert codes.shape == (2, cfg.num_quantizers, ecfg.frames(L)), (codes.shape, ecfg.frames(L))
            got_codes, got_emb = eo.encode(ecfg, cfg, w, audio)
            assert torch.allclose(got_emb, emb, atol=2e-5, rtol=1e-4), float((got_emb - emb).abs().max())
            agree = float((got_codes == codes).float().mean())
            assert agree >= 0.99, agree                      # a near-tie may fall either way under another summation order
            arrays[f"audio_{L}"] = audio
            arrays[f"emb_{L}"] = emb
            arrays[f"codes_{L}"] = codes
        save(tag, dict(encoder_config=ecfg.to_dict(), config=cfg.to_dict(), weight_seeds=[17, 19], weight_abs_sum=wsum(w), lengths=list(lengths)),
             **arrays)


# ---------------------------------------------------------------------------------------------
def make_primitives():
    g = torch.Generator().manual_seed(21)
    arrays, meta = {}, {"cases": []}

    def rnd(*shape, s=1.0):
        return torch.randn(*shape, generator=g) * s

    # causal conv: every (k, dilation) the decoder uses, plus depthwise
    for k, d, cin, cout, groups, L in ((7, 1, 16, 16, 1, 100), (7, 3, 16, 16, 1, 100), (7, 9, 8, 8, 1, 130),
                                       (1, 1, 16, 16, 1, 50), (3, 1, 32, 16, 1, 40), (7, 1, 16, 16, 16, 60),
                                       (7, 1, 32, 1, 1, 257), (7, 9, 8, 8, 1, 5)):
        m = sib.Qwen3OmniMoeCausalConvNet(cin, cout, k, dilation=d, groups=groups)
        with torch.no_grad():
            m.conv.weight.copy_(rnd(*m.conv.weight.shape, s=0.3)); m.conv.bias.copy_(rnd(cout, s=0.1))
            x = rnd(2, cin, L)
            y = m(x)
        mine = oracle.causal_conv1d(x, m.conv.weight.detach(), m.conv.bias.detach(), d, groups)
        assert torch.allclose(mine, y, atol=1e-6), (k, d)
        tag = f"cconv_k{k}_d{d}_g{groups}_c{cin}x{cout}_L{L}"
        meta["cases"].append(dict(kind="causal_conv", tag=tag, k=k, dilation=d, groups=groups))
        arrays.update({tag + ".x": x, tag + ".w": m.conv.weight.detach(), tag + ".b": m.conv.bias.detach(), tag + ".y": y})
    # transposed conv: every (k, s) used
    for k, s, cin, cout, L in ((2, 2, 16, 16, 30), (16, 8, 16, 8, 20), (10, 5, 16, 8, 20), (8, 4, 16, 8, 33),
                               (6, 3, 16, 8, 41), (16, 8, 8, 8, 1), (6, 3, 8, 8, 2)):
        m = sib.Qwen3OmniMoeCausalTransConvNet(cin, cout, k, s)
        with torch.no_grad():
            m.conv.weight.copy_(rnd(*m.conv.weight.shape, s=0.3)); m.conv.bias.copy_(rnd(cout, s=0.1))
            x = rnd(2, cin, L)
            y = m(x)
        mine = oracle.causal_trans_conv1d(x, m.conv.weight.detach(), m.conv.bias.detach(), s)
        assert mine.shape == y.shape and torch.allclose(mine, y, atol=1e-6), (k, s)
        tag = f"tconv_k{k}_s{s}_c{cin}x{cout}_L{L}"
        meta["cases"].append(dict(kind="trans_conv", tag=tag, k=k, stride=s))
        arrays.update({tag + ".x": x, tag + ".w": m.conv.weight.detach(), tag + ".b": m.conv.bias.detach(), tag + ".y": y})
    # snake
    for C, L in ((8, 64), (32, 33)):
        m = sib.Qwen3OmniMoeSnakeBeta(C)
        with torch.no_grad():
            m.alpha.copy_(rnd(C, s=0.3)); m.beta.copy_(rnd(C, s=0.3))
            x = rnd(2, C, L, s=2.0)
            y = m(x)
        mine = oracle.snake_beta(x, m.alpha.detach(), m.beta.detach())
        assert torch.allclose(mine, y, atol=1e-6)
        tag = f"snake_c{C}_L{L}"
        meta["cases"].append(dict(kind="snake", tag=tag))
        arrays.update({tag + ".x": x, tag + ".alpha": m.alpha.detach(), tag + ".beta": m.beta.detach(), tag + ".y": y})
    # ConvNeXt block
    for C, L in ((16, 50), (32, 9)):
        m = sib.Qwen3OmniMoeConvNeXtBlock(C)
        w = {}
        with torch.no_grad():
            for n, p in m.named_parameters():
                p.copy_(rnd(*p.shape, s=0.3) if p.dim() > 1 else (0.5 + rnd(*p.shape, s=0.1)))
                w["cn." + n] = p.detach().clone()
            x = rnd(2, C, L)
            y = m(x)
        mine = oracle.convnext_block(x, w, "cn.")
        assert torch.allclose(mine, y, atol=2e-6), (mine - y).abs().max()
        tag = f"convnext_c{C}_L{L}"
        meta["cases"].append(dict(kind="convnext", tag=tag, params=sorted(w)))
        arrays.update({tag + ".x": x, tag + ".y": y, **{tag + "." + k: v for k, v in w.items()}})
    # residual unit and decoder block
    for C, d, L in ((16, 1, 80), (16, 3, 80), (8, 9, 120)):
        m = sib.Qwen3OmniMoeCode2WavDecoderResidualUnit(C, d)
        w = {}
        with torch.no_grad():
            for n, p in m.named_parameters():
                p.copy_(rnd(*p.shape, s=0.3))
                w["ru." + n] = p.detach().clone()
            x = rnd(2, C, L)
            y = m(x)
        mine = oracle.residual_unit(x, w, "ru.", d)
        assert torch.allclose(mine, y, atol=2e-6)
        tag = f"resunit_c{C}_d{d}_L{L}"
        meta["cases"].append(dict(kind="residual_unit", tag=tag, dilation=d, params=sorted(w)))
        arrays.update({tag + ".x": x, tag + ".y": y, **{tag + "." + k: v for k, v in w.items()}})
    save("primitives", meta, **arrays)


# ---------------------------------------------------------------------------------------------
def sibling_model(cfg, w):
    sc = Qwen3OmniMoeCode2WavConfig(
        codebook_size=cfg.codebook_size, hidden_size=cfg.hidden_size, num_attention_heads=cfg.num_attention_heads,
        num_key_value_heads=cfg.num_key_value_heads, sliding_window=cfg.sliding_window,
        intermediate_size=cfg.intermediate_size, rms_norm_eps=cfg.rms_norm_eps, num_hidden_layers=cfg.num_hidden_layers,
        num_quantizers=cfg.num_quantizers, upsample_rates=list(cfg.upsample_rates),
        upsampling_ratios=list(cfg.upsampling_ratios), decoder_dim=cfg.decoder_dim)
    sc._attn_implementation = "eager"
    m = sib.Qwen3OmniMoeCode2Wav(sc).eval().float()
    missing, unexpected = m.load_state_dict(w, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in k or "code_offset" in k or "inv_freq" in k for k in missing), missing
    return m


def make_decoder():
    for tag, over in (("decoder_tiny", {}), ("decoder_gqa", dict(num_key_value_heads=2, sliding_window=8))):
        cfg = q3.tiny_config(frontend=q3.FRONTEND_MEAN_EMBEDDING, **over)
        assert cfg.head_dim * cfg.num_attention_heads == cfg.hidden_size
        w = q3.synthetic_weights(cfg, seed=3)
        m = sibling_model(cfg, w)
        arrays = {}
        meta = dict(config=cfg.to_dict(), weight_seed=3, weight_abs_sum=wsum(w), cases=[])
        for T in ((1, 2, 40, 73) if tag == "decoder_tiny" else (1, 2, 40)):
            codes = q3.random_codes(cfg, 2, T, seed=100 + T)
            with torch.no_grad():
                want = m(codes)
            got = oracle.forward(cfg, w, codes)
            assert got.shape == want.shape == (2, 1, cfg.output_length(T)), (got.shape, want.shape)
            err = (got - want).abs().max().item()
            print(f"{tag} T={T} len={want.shape[-1]} max|oracle-sibling|={err:.3e} rms={want.pow(2).mean().sqrt():.3f} "
                  f"clamped={(want.abs() >= 1).float().mean():.4f}")
            assert err < 2e-5, err
            arrays[f"T{T}.codes"] = codes
            arrays[f"T{T}.wav"] = want[:, 0] if T < 40 else want[1:, 0]      # long cases keep the LAST item only
            meta["cases"].append(dict(T=T, length=int(want.shape[-1]), items_stored=2 if T < 40 else 1))
        if tag == "decoder_tiny":
            # chunk edge: 310 frames = chunks [0,300) and [275,310); store a strided sample of the 594 090 values
            T = 310
            codes = q3.random_codes(cfg, 1, T, seed=100 + T)
            with torch.no_grad():
                want = m.chunked_decode(codes, chunk_size=300, left_context_size=25)
            got = oracle.chunked_decode(cfg, w, codes, 300, 25)
            assert got.shape == want.shape == (1, 1, cfg.chunked_output_length(T))
            err = (got - want).abs().max().item()
            print(f"{tag} chunked T={T} len={want.shape[-1]} max err {err:.3e}")
            assert err < 2e-5
            seam = 300 * cfg.total_upsample - cfg.trim
            arrays["T310.codes"] = codes
            arrays["T310.wav_stride7"] = want[0, 0, 3::7]
            arrays["T310.wav_seam"] = want[0, 0, seam - 2048: seam + 2048]
            meta["cases"].append(dict(T=T, length=int(want.shape[-1]), chunked=True, seam=int(seam)))
            # perturbation fact: changing frame t changes samples from t*1920 - 555 onward only
            T, t = 12, 7
            codes = q3.random_codes(cfg, 1, T, seed=77)
            codes2 = codes.clone(); codes2[0, :, t] = (codes2[0, :, t] + 1) % cfg.codebook_size
            with torch.no_grad():
                a, b = m(codes), m(codes2)
            first = int((a != b).nonzero()[:, -1].min())
            assert first >= t * cfg.total_upsample - cfg.trim, first
            meta["perturbation"] = dict(T=T, t=t, first_changed=first, bound=t * cfg.total_upsample - cfg.trim)
        save(tag, meta, **arrays)


def make_transformer_parts():
    """Attention/RoPE/mask in isolation at the window edge, from the sibling transformer model."""
    cfg = q3.tiny_config(frontend=q3.FRONTEND_MEAN_EMBEDDING, sliding_window=5)
    w = q3.synthetic_weights(cfg, seed=9)
    m = sibling_model(cfg, w)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 19, cfg.hidden_size, generator=g)
    with torch.no_grad():
        want = m.pre_transformer(inputs_embeds=x).last_hidden_state
    got = oracle.transformer(cfg, x, w)
    err = (got - want).abs().max().item()
    print(f"transformer window=5 max err {err:.3e}")
    assert err < 2e-5
    save("transformer_w5", dict(config=cfg.to_dict(), weight_seed=9, weight_abs_sum=wsum(w)), x=x, y=want)


if __name__ == "__main__":
    if len(sys.argv) > 1:                 # only the named fixtures, e.g. `make_goldens.py make_encoder`
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    make_rvq()
    make_rvq_encode()
    make_encoder()
    make_primitives()
    make_decoder()
    make_transformer_parts()
