"""The CPU oracle against the golden fixtures (sibling outputs).  CPU only."""
import numpy as np
import pytest
import torch

import qwen3_tts_amd as q3
import decoder_oracle as oracle


def _weights(meta):
    cfg = q3.DecoderConfig.from_dict(meta["config"])
    w = q3.synthetic_weights(cfg, seed=meta["weight_seed"])
    s = float(sum(t.double().abs().sum().item() for t in w.values()))
    assert abs(s - meta["weight_abs_sum"]) <= 1e-9 * abs(s), "seeded weight generator drifted from the fixture"
    return cfg, w


@pytest.mark.parametrize("name", ["rvq_tiny", "rvq_q16", "rvq_sem2"])
def test_rvq_bit_exact(golden, name):
    meta, a = golden(name)
    cfg, w = _weights(meta)
    codes = torch.from_numpy(a["codes"])
    out, sums = oracle.rvq_dequant(cfg, w, codes)
    assert meta["clamped_rows"] > 0
    got = torch.stack(sums, 1).numpy()
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), a["group_sums"].view(np.uint32))
    np.testing.assert_allclose(out.numpy(), a["out"], atol=1e-5, rtol=1e-5)


def _encode_weights(cfg, w, a):
    extra = {k.replace("__", "."): torch.from_numpy(v) for k, v in a.items() if k.startswith("quantizer__")}
    assert set(extra) == {"quantizer.semantic.input_proj.weight", "quantizer.acoustic.input_proj.weight"}
    return {**w, **extra}


@pytest.mark.parametrize("name", ["rvq_encode_tiny", "rvq_encode_q16"])
def test_rvq_encode_equals_the_sibling(golden, name):
    """The encode direction (input_proj, nearest row per layer, residual passed on; both groups on the same embeddings)
    against mimi's split RVQ `encode`: identical codes."""
    meta, a = golden(name)
    cfg, w = _weights(meta)
    got = oracle.rvq_encode(cfg, _encode_weights(cfg, w, a), torch.from_numpy(a["embeddings"]))
    assert got.dtype == torch.int64 and torch.equal(got, torch.from_numpy(a["codes"]))
    # the codes are a greedy residual quantisation: decoding them approaches the projected embedding, layer by layer
    ws = _encode_weights(cfg, w, a)
    x = torch.nn.functional.conv1d(torch.from_numpy(a["embeddings"]), ws["quantizer.acoustic.input_proj.weight"]).permute(0, 2, 1)
    ns = cfg.num_semantic_quantizers
    res, norms = x.clone(), []
    for i in range(cfg.num_quantizers - ns):
        tab = oracle.codebook_table(w[f"quantizer.acoustic.layers.{i}.embed_sum"], w[f"quantizer.acoustic.layers.{i}.cluster_usage"], cfg.rvq_eps)
        res = res - tab[got[:, ns + i]]
        norms.append(float(res.norm()))
    assert all(b <= a_ + 1e-6 for a_, b in zip([float(x.norm())] + norms[:-1], norms))


def encoder_fixture(meta):
    """Configurations and seeded weights of an encoder fixture (shared with the GPU test of the encode side)."""
    import importlib
    E = importlib.import_module("qwen3-tts_amd.encoder")
    ecfg = E.EncoderConfig.from_dict(meta["encoder_config"])
    cfg = q3.DecoderConfig.from_dict(meta["config"])
    w = {**q3.synthetic_weights(cfg, seed=meta["weight_seeds"][0]), **E.synthetic_encoder_weights(ecfg, cfg.codebook_dim, seed=meta["weight_seeds"][1])}
    s = float(sum(t.double().abs().sum().item() for t in w.values()))
    assert abs(s - meta["weight_abs_sum"]) <= 1e-9 * abs(s), "seeded weight generator drifted from the fixture"
    return ecfg, cfg, w


@pytest.mark.parametrize("name", ["encoder_tiny", "encoder_wide"])
def test_encode_side_equals_the_sibling(golden, name):
    """The tokenizer's encode side (conv encoder with ELU residual blocks and strided convs, LayerNorm transformer, the
    replicate-padded stride-2 conv, split-RVQ encode) against the sibling MimiModel's stored outputs: the embeddings it
    quantises to fp32 rounding, its codes up to near-ties (a frame whose two nearest rows are closer than the summation
    noise may fall either way; everything after it in that frame's residual chain then differs), at lengths that no stride
    divides and one shorter than a frame."""
    import encoder_oracle as eo
    meta, a = golden(name)
    ecfg, cfg, w = encoder_fixture(meta)
    for L in meta["lengths"]:
        audio = torch.from_numpy(a[f"audio_{L}"])
        codes, emb = eo.encode(ecfg, cfg, w, audio)
        assert codes.shape == (audio.shape[0], cfg.num_quantizers, ecfg.frames(L)) and codes.dtype == torch.int64
        np.testing.assert_allclose(emb.numpy(), a[f"emb_{L}"], atol=2e-5, rtol=1e-4)
        assert float((codes.numpy() == a[f"codes_{L}"]).mean()) >= 0.99


def test_rvq_negative_zero_start():
    """The accumulator starts at +0.0, so a single row of -0.0 dequantises to +0.0."""
    table = torch.tensor([[-0.0, 1.0]], dtype=torch.float32)
    out = oracle.rvq_group_sum(torch.zeros(1, 1, 1, dtype=torch.int64), [table])
    assert out.view(torch.int32)[0, 0, 0].item() == 0


def test_primitives(golden):
    meta, a = golden("primitives")
    T = lambda k: torch.from_numpy(a[k])
    for c in meta["cases"]:
        tag = c["tag"]
        if c["kind"] == "causal_conv":
            y = oracle.causal_conv1d(T(tag + ".x"), T(tag + ".w"), T(tag + ".b"), c["dilation"], c["groups"])
        elif c["kind"] == "trans_conv":
            y = oracle.causal_trans_conv1d(T(tag + ".x"), T(tag + ".w"), T(tag + ".b"), c["stride"])
        elif c["kind"] == "snake":
            y = oracle.snake_beta(T(tag + ".x"), T(tag + ".alpha"), T(tag + ".beta"))
        elif c["kind"] == "convnext":
            w = {p: T(tag + "." + p) for p in c["params"]}
            y = oracle.convnext_block(T(tag + ".x"), w, "cn.")
        elif c["kind"] == "residual_unit":
            w = {p: T(tag + "." + p) for p in c["params"]}
            y = oracle.residual_unit(T(tag + ".x"), w, "ru.", c["dilation"])
        else:
            raise AssertionError(c)
        assert tuple(y.shape) == a[tag + ".y"].shape, tag
        np.testing.assert_allclose(y.numpy(), a[tag + ".y"], atol=3e-6, rtol=1e-5, err_msg=tag)


def test_transformer_window_edge(golden):
    meta, a = golden("transformer_w5")
    cfg, w = _weights(meta)
    y = oracle.transformer(cfg, torch.from_numpy(a["x"]), w)
    np.testing.assert_allclose(y.numpy(), a["y"], atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("name", ["decoder_tiny", "decoder_gqa"])
def test_decoder_end_to_end(golden, name):
    meta, a = golden(name)
    cfg, w = _weights(meta)
    for c in meta["cases"]:
        T = c["T"]
        codes = torch.from_numpy(a[f"T{T}.codes"])
        if c.get("chunked"):
            wav = oracle.chunked_decode(cfg, w, codes, 300, 25)[0, 0].numpy()
            assert wav.shape[0] == c["length"] == cfg.chunked_output_length(T)
            np.testing.assert_allclose(wav[3::7], a[f"T{T}.wav_stride7"], atol=2e-5)
            s = c["seam"]
            np.testing.assert_allclose(wav[s - 2048: s + 2048], a[f"T{T}.wav_seam"], atol=2e-5)
        else:
            wav = oracle.forward(cfg, w, codes)[:, 0].numpy()
            assert wav.shape[1] == c["length"] == cfg.output_length(T)
            want = a[f"T{T}.wav"]
            np.testing.assert_allclose(wav[-want.shape[0]:], want, atol=2e-5)
            assert np.abs(wav).max() <= 1.0


def test_decoder_properties(golden):
    meta, _ = golden("decoder_tiny")
    cfg, w = _weights(meta)
    # lengths
    for T in (1, 3, 299, 300, 301, 325, 326, 600, 601):
        n = -(-T // 300)
        assert cfg.chunked_output_length(T) == T * 1920 - 555 * n
    # prefix property: decode(codes[:T']) == decode(codes)[:len']
    codes = q3.random_codes(cfg, 1, 9, seed=1)
    full = oracle.forward(cfg, w, codes)
    part = oracle.forward(cfg, w, codes[..., :5])
    np.testing.assert_allclose(full[..., : part.shape[-1]].numpy(), part.numpy(), atol=1e-4)  # fp32 blocking noise ~2e-5
    # lookahead: frame t first influences sample t*1920 - 555
    p = meta["perturbation"]
    c1 = q3.random_codes(cfg, 1, p["T"], seed=77)
    c2 = c1.clone(); c2[0, :, p["t"]] = (c2[0, :, p["t"]] + 1) % cfg.codebook_size
    a, b = oracle.forward(cfg, w, c1), oracle.forward(cfg, w, c2)
    first = int((a != b).nonzero()[:, -1].min())
    assert first >= p["bound"] and first == p["first_changed"]


def test_wrong_codebook_count_raises(golden):
    meta, _ = golden("decoder_tiny")
    cfg, w = _weights(meta)
    with pytest.raises(ValueError):
        oracle.forward(cfg, w, torch.zeros(1, cfg.num_quantizers + 1, 4, dtype=torch.int64))


def test_decode_wrapper_returns_numpy_list(golden):
    meta, _ = golden("decoder_tiny")
    cfg, w = _weights(meta)
    items = [q3.random_codes(cfg, 1, T, seed=T)[0].t().contiguous() for T in (2, 5)]
    wavs, sr = oracle.decode(cfg, w, items)
    assert sr == 24000 and [x.shape for x in wavs] == [(2 * 1920 - 555,), (5 * 1920 - 555,)]
    assert all(x.dtype == np.float32 for x in wavs)


def test_pcm16_oracle_known_answers_and_wav_container():
    """Float -> PCM_16 rule (lrintf(x * 32767), ties to even) on hand-checked values, and the WAV container against
    scipy's writer (the conversion itself is unpinned: soundfile is not in the container)."""
    import io
    import pcm16_oracle as po
    from scipy.io import wavfile
    x = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 0.5 / 32767, 1.5 / 32767, 2.5 / 32767, -0.5 / 32767, -1.5 / 32767, 1e-9, 2.0, -2.0],
                 dtype=np.float32)
    want = np.array([0, 0, 32767, -32767, 16384, 0, 2, 2, 0, -2, 0, 32767, -32768], dtype=np.int16)
    assert np.array_equal(po.float_to_pcm16(x), want)                 # 0.5 * 32767 = 16383.5 -> 16384 (even)
    pcm = po.float_to_pcm16(np.sin(np.arange(1000, dtype=np.float32) * 0.01))
    buf = io.BytesIO()
    wavfile.write(buf, 24000, pcm)
    assert po.wav_file_bytes(pcm, 24000) == buf.getvalue()

