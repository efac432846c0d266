// The tokenizer's ENCODE side (24 kHz audio -> 12.5 Hz code frames): weight packing and the launch sequence.  Part of
// decoder.cpp's translation unit (included at its end): it shares the context, the packing helpers and gemm().
// Arithmetic: the sibling MimiModel.encode (transformers/models/mimi/modeling_mimi.py:1230-1263): conv encoder :450-492 with
// residual blocks :408-447, transformer :729-779, stride-2 conv :1197-1206, split-RVQ encode :1099-1126.
//
// Data flow, channels-last [B][L][C] in the context's element type.  ELU sits in front of every conv but the first; the
// producing launch writes it, as its activation or as a second output beside the raw tensor that a skip path needs, so no
// elementwise launch remains.  A strided conv (kernel 2r, stride r, causal) over rows padded to a whole number
// of strides is a two-tap contraction over the rows viewed as [L / r][r * C]:
//   y[m] = W[:, :, 0:r] . x'[m - 1] + W[:, :, r:2r] . x'[m]
// which is the decode side's transposed-conv trick run backwards.
namespace {

// ConvTranspose-free strided Conv1d weight [Cout][Cin][2r] -> two taps of [Cout][r * Cin]
int pack_strided(q3c_ctx* c, const std::string& prefix, int cout, int cin, int r, int dtype, bool has_bias, int off0, GemmW* g) {
    int rc = 0;
    GETW(w, prefix + ".weight", cout, cin, 2 * r);
    std::vector<float> p((size_t)2 * cout * r * cin);
    for (int t = 0; t < 2; ++t)
        for (int n = 0; n < cout; ++n)
            for (int j = 0; j < r; ++j)
                for (int ci = 0; ci < cin; ++ci)
                    p[(((size_t)t * cout + n) * r + j) * cin + ci] = w->data[((size_t)n * cin + ci) * 2 * r + t * r + j];
    g->taps = 2; g->N = cout; g->K = r * cin; g->off0 = off0; g->off_step = 1; g->in_dtype = dtype;
    if ((rc = upload_typed(c, p, dtype, &g->w))) return rc;
    if (has_bias) {
        GETW(b, prefix + ".bias", cout);
        if ((rc = upload_f32(c, b->data, &g->bias))) return rc;
    }
    return 0;
}

int encoder_finalize(q3c_ctx* c) {
    EncoderState& e = c->enc;
    const q3c_encoder_config& k = e.cfg;
    const int dt = c->dtype;
    int rc = 0;
    if (c->cfg.frontend != Q3C_FRONTEND_SPLIT_RVQ) return fail(c, Q3C_ERR_STATE, "the encode side needs the split-RVQ front end");
    if (!c->rvq_in_proj || c->rvq_in_dim != k.hidden_size)
        return fail(c, Q3C_ERR_WEIGHT, "the encode side needs quantizer.{semantic,acoustic}.input_proj.weight of shape [codebook_dim][encoder hidden_size][1]");
    {
        GETW(w0, "encoder.layers.0.conv.weight", k.num_filters, 1, k.kernel_size);
        TRY(upload_f32(c, w0->data, &e.conv0_w));
        GETW(b0, "encoder.layers.0.conv.bias", k.num_filters);
        TRY(upload_f32(c, b0->data, &e.conv0_b));
    }
    int dim = k.num_filters, idx = 1;
    e.stages.resize(k.num_ratios);
    for (int i = 0; i < k.num_ratios; ++i) {
        EncStage& st = e.stages[i];
        st.r = k.ratios[i]; st.dim = dim;
        const std::string p = "encoder.layers." + std::to_string(idx) + ".block.";
        TRY(pack_conv(c, p + "1.conv", dim / k.compress, dim, k.residual_kernel_size, 1, dt, &st.c3));
        TRY(pack_conv(c, p + "3.conv", dim, dim / k.compress, 1, 1, dt, &st.c1));
        TRY(pack_strided(c, "encoder.layers." + std::to_string(idx + 2) + ".conv", 2 * dim, dim, st.r, dt, true, -1, &st.down));
        dim *= 2;
        idx += 3;
    }
    TRY(pack_conv(c, "encoder.layers." + std::to_string(idx + 1) + ".conv", k.hidden_size, dim, k.last_kernel_size, 1, dt, &e.last));
    const int H = k.hidden_size, I = k.intermediate_size;
    e.layers.resize(k.num_hidden_layers);
    for (int l = 0; l < k.num_hidden_layers; ++l) {
        EncLayer& y = e.layers[l];
        const std::string p = "encoder_transformer.layers." + std::to_string(l) + ".";
        TRY(pack_vec(c, p + "input_layernorm.weight", H, &y.ln1_w));
        TRY(pack_vec(c, p + "input_layernorm.bias", H, &y.ln1_b));
        TRY(pack_vec(c, p + "post_attention_layernorm.weight", H, &y.ln2_w));
        TRY(pack_vec(c, p + "post_attention_layernorm.bias", H, &y.ln2_b));
        TRY(pack_vec(c, p + "self_attn_layer_scale.scale", H, &y.ls_attn));
        TRY(pack_vec(c, p + "mlp_layer_scale.scale", H, &y.ls_mlp));
        GETW(wq, p + "self_attn.q_proj.weight", H, H);
        GETW(wk, p + "self_attn.k_proj.weight", H, H);
        GETW(wv, p + "self_attn.v_proj.weight", H, H);
        std::vector<float> cat;
        cat.reserve((size_t)3 * H * H);
        cat.insert(cat.end(), wq->data.begin(), wq->data.end());
        cat.insert(cat.end(), wk->data.begin(), wk->data.end());
        cat.insert(cat.end(), wv->data.begin(), wv->data.end());
        y.qkv.taps = 1; y.qkv.N = 3 * H; y.qkv.K = H; y.qkv.in_dtype = dt;
        TRY(upload_typed(c, cat, dt, &y.qkv.w));
        TRY(pack_linear(c, p + "self_attn.o_proj.weight", "", H, H, dt, &y.o));
        TRY(pack_linear(c, p + "mlp.fc1.weight", "", I, H, dt, &y.fc1));
        TRY(pack_linear(c, p + "mlp.fc2.weight", "", H, I, dt, &y.fc2));
    }
    // the stride-2 conv reads a replicate-padded copy whose first group is the padding: taps at rows m and m + 1
    TRY(pack_strided(c, "downsample.conv", H, H, k.downsample_stride, dt, false, 0, &e.ds));
    {   // rotary tables, as for the decode side's transformer
        const int half = k.head_dim / 2;
        GETW(inv, "encoder_rope.inv_freq", half);
        std::vector<float> ct((size_t)k.max_positions * half), st(ct.size());
        for (int pos = 0; pos < k.max_positions; ++pos)
            for (int j = 0; j < half; ++j) {
                const float ang = (float)pos * inv->data[j];
                ct[(size_t)pos * half + j] = (float)std::cos((double)ang);
                st[(size_t)pos * half + j] = (float)std::sin((double)ang);
            }
        TRY(upload_f32(c, ct, &e.cos_t));
        TRY(upload_f32(c, st, &e.sin_t));
    }
    e.ready = true;
    return 0;
}

long long enc_frames(const q3c_encoder_config& k, long long samples, long long* at_transformer) {
    long long n = samples;
    for (int i = 0; i < k.num_ratios; ++i) n = (n + k.ratios[i] - 1) / k.ratios[i];
    if (at_transformer) *at_transformer = n;
    return (n + k.downsample_stride - 1) / k.downsample_stride;
}

// The launch sequence (run) or only its workspace size (!run).
int encode_sequence(q3c_ctx* c, int B, int L0, bool run, const float* audio, int64_t* codes, float* emb_out, char* ws, hipStream_t s,
                    size_t* ws_total) {
    const EncoderState& e = c->enc;
    const q3c_encoder_config& k = e.cfg;
    const int dt = c->dtype;
    const size_t es = esize(dt);
    size_t off = 0;
    auto alloc = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const uintptr_t ws_base = (uintptr_t)ws;       // the layout pass has no workspace: integer offsets, never null-pointer arithmetic
    auto at = [&](size_t o) { return (void*)(ws_base + o); };

    // x: a stage's raw input (its residual block's skip path); e: ELU(x), what the block's first conv reads, in the layout
    // of the stage's strided conv (items padded to whole strides, the padding rows zero).  Each producer writes both.
    int L = L0, dim = k.num_filters;
    size_t o_x = alloc((size_t)B * L * dim * es);
    auto padded = [&](int len, size_t i) { const int r = i < e.stages.size() ? e.stages[i].r : 1; return (len + r - 1) / r * r; };
    int Lp = padded(L, 0);
    size_t o_e = alloc((size_t)B * Lp * dim * es);
    if (run) {
        if (Lp != L) HIPCHK(c, hipMemsetAsync(at(o_e), 0, (size_t)B * Lp * dim * es, s));
        LAUNCH_B(c, s, "audio_conv", (size_t)B * L * (4 + 2 * dim * es),
                 q3_launch_audio_conv(audio, e.conv0_w, e.conv0_b, at(o_x), at(o_e), (long long)Lp * dim, B, L, dim, k.kernel_size, dt, s));
    }
    for (size_t si = 0; si < e.stages.size(); ++si) {
        const EncStage& st = e.stages[si];
        const int hid = dim / k.compress, r = st.r, Ln = Lp / r;
        const size_t o_h = alloc((size_t)B * L * hid * es), o_y = alloc((size_t)B * Ln * 2 * dim * es);
        const int Lpn = padded(Ln, si + 1);
        const size_t o_en = alloc((size_t)B * Lpn * 2 * dim * es);
        if (run) {
            // the block: h = ELU(conv3(e)); e <- ELU(x + conv1(h)) (x itself has no other reader); then the strided conv on e
            // (rows viewed as [Lp / r][r * dim]) writes the next stage's x and e
            GemmOpts g3; g3.act = Q3_ACT_ELU; g3.x_bstride = (long long)Lp * dim;
            TRY(gemm(c, st.c3, at(o_e), B, L, at(o_h), L, dt, g3, s));   // reads the first L rows of each item of e
            GemmOpts g1; g1.R = at(o_x); g1.post_elu = true; g1.y_bstride = (long long)Lp * dim;
            TRY(gemm(c, st.c1, at(o_h), B, L, at(o_e), L, dt, g1, s));
            if (Lpn != Ln) HIPCHK(c, hipMemsetAsync(at(o_en), 0, (size_t)B * Lpn * 2 * dim * es, s));
            GemmOpts gd; gd.post_elu = true;
            if (si + 1 < e.stages.size()) {
                gd.Y2 = at(o_en); gd.y2_bstride = (long long)Lpn * 2 * dim;
                TRY(gemm(c, st.down, at(o_e), B, Ln, at(o_y), Ln, dt, gd, s));
            } else {
                TRY(gemm(c, st.down, at(o_e), B, Ln, at(o_en), Ln, dt, gd, s));      // only the last conv reads this: ELU alone
            }
        }
        o_x = o_y; o_e = o_en; L = Ln; Lp = Lpn; dim *= 2;
    }
    const int H = k.hidden_size, I = k.intermediate_size, T = L;       // T: rows at the transformer's rate
    if (T > k.max_positions) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "audio longer than the encoder's max_positions");
    const size_t o_z = alloc((size_t)B * T * H * es), o_n = alloc((size_t)B * T * H * es);
    const size_t o_qkv = alloc((size_t)B * T * 3 * H * es), o_att = alloc((size_t)B * T * H * es), o_mlp = alloc((size_t)B * T * I * es);
    const int sd = k.downsample_stride, Tg = (T + sd - 1) / sd;
    const size_t o_pad = alloc((size_t)B * (Tg + 1) * sd * H * es), o_emb = alloc((size_t)B * Tg * H * 4);
    if (run) {
        TRY(gemm(c, e.last, at(o_e), B, T, at(o_z), T, dt, GemmOpts{}, s));
        void* x = at(o_z); void* nrm = at(o_n);
        const long long rows = (long long)B * T;
        for (const EncLayer& y : e.layers) {
            LAUNCH_B(c, s, "layernorm", 2.0 * rows * H * es, q3_launch_layernorm(x, y.ln1_w, y.ln1_b, nrm, rows, H, k.norm_eps, dt, s));
            TRY(gemm(c, y.qkv, nrm, B, T, at(o_qkv), T, dt, GemmOpts{}, s));
            LAUNCH_B(c, s, "swa_attention", 4.0 * rows * H * es,
                     q3_launch_swa_attention(at(o_qkv), at(o_att), e.cos_t, e.sin_t, B, T, k.num_attention_heads, k.num_attention_heads, k.head_dim,
                                             k.sliding_window, dt, s, 0, nullptr, k.max_positions, nullptr, c->tuning));
            GemmOpts o1; o1.scale = y.ls_attn; o1.R = x;
            TRY(gemm(c, y.o, at(o_att), B, T, x, T, dt, o1, s));
            LAUNCH_B(c, s, "layernorm", 2.0 * rows * H * es, q3_launch_layernorm(x, y.ln2_w, y.ln2_b, nrm, rows, H, k.norm_eps, dt, s));
            GemmOpts o2; o2.act = Q3_ACT_GELU;
            TRY(gemm(c, y.fc1, nrm, B, T, at(o_mlp), T, dt, o2, s));
            GemmOpts o3; o3.scale = y.ls_mlp; o3.R = x;
            TRY(gemm(c, y.fc2, at(o_mlp), B, T, x, T, dt, o3, s));
        }
        // stride-2 conv, replicate padding: sd copies of the first row in front, the last row repeated up to a whole stride
        LAUNCH_B(c, s, "replicate_pad", 2.0 * rows * H * es, q3_launch_replicate_pad(x, at(o_pad), B, T, H, sd, (Tg + 1) * sd, dt, s));
        float* emb = emb_out ? emb_out : (float*)at(o_emb);
        TRY(gemm(c, e.ds, at(o_pad), B, Tg + 1, emb, Tg, Q3_F32, GemmOpts{}, s));
        LAUNCH(c, q3_launch_rvq_encode(emb, B, Tg, c->rvq_in_dim, c->cfg.num_quantizers, c->cfg.codebook_size, c->cfg.codebook_dim, c->tables_dev,
                                       (const void* const*)c->tables_t_dev, c->group_start_dev, c->G, c->rvq_in_proj, (long long*)codes, s, c->tuning));
    }
    if (ws_total) *ws_total = off;
    return 0;
}

}  // namespace

extern "C" {

int q3c_encoder_configure(q3c_ctx* c, const q3c_encoder_config* k) {
    if (!c || !k) return Q3C_ERR_INVALID_ARGUMENT;
    if (c->finalized) return fail(c, Q3C_ERR_STATE, "q3c_encoder_configure must precede q3c_finalize");
    auto bad = [&](const char* m) { return fail(c, Q3C_ERR_INVALID_ARGUMENT, m); };
    if (k->struct_size != (int)sizeof(q3c_encoder_config)) return bad("q3c_encoder_config.struct_size mismatch");
    if (k->num_ratios < 1 || k->num_ratios > Q3C_MAX_STAGES) return bad("encoder: stage count out of range");
    for (int i = 0; i < k->num_ratios; ++i) if (k->ratios[i] < 1) return bad("encoder: ratios must be >= 1");
    if (k->num_filters < 8 || k->kernel_size < 1 || k->residual_kernel_size < 1 || k->last_kernel_size < 1 || k->compress < 1 ||
        k->hidden_size < 8 || k->num_attention_heads < 1 || k->head_dim < 2 || k->intermediate_size < 8 || k->num_hidden_layers < 0 ||
        k->downsample_stride < 1 || k->max_positions < 1)
        return bad("encoder: sizes must be positive");
    if (k->num_filters % (8 * k->compress) || k->hidden_size % 8 || k->intermediate_size % 8 || (k->head_dim & 1))
        return fail(c, Q3C_ERR_UNSUPPORTED, "encoder: channel counts must be multiples of 8 (num_filters of 8 x compress), head_dim even");
    if (k->hidden_size != k->num_attention_heads * k->head_dim) return bad("encoder: hidden_size != heads x head_dim");
    if (k->sliding_window < 1 || k->sliding_window > 253) return fail(c, Q3C_ERR_UNSUPPORTED, "encoder: sliding_window must be in [1, 253]");
    c->enc.cfg = *k;
    c->enc.configured = true;
    return 0;
}

int q3c_encode_frames(const q3c_ctx* c, int64_t samples, int64_t* frames) {
    if (!c || !frames || samples < 1 || !c->enc.configured) return Q3C_ERR_INVALID_ARGUMENT;
    *frames = enc_frames(c->enc.cfg, samples, nullptr);
    return 0;
}

int q3c_encode_workspace_bytes(q3c_ctx* c, int32_t B, int32_t samples, size_t* bytes) {
    if (!c || !bytes) return Q3C_ERR_INVALID_ARGUMENT;
    if (!c->finalized || !c->enc.ready) return fail(c, Q3C_ERR_STATE, "no finalized encoder in this context");
    if (B < 1 || samples < 1) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "batch and samples must be >= 1");
    return encode_sequence(c, B, samples, false, nullptr, nullptr, nullptr, nullptr, nullptr, bytes);
}

int q3c_encode(q3c_ctx* c, const float* audio, int32_t B, int32_t samples, int64_t* codes, float* embeddings, void* ws, size_t ws_bytes,
               void* stream) {
    if (!c) return Q3C_ERR_INVALID_ARGUMENT;
    if (!c->finalized || !c->enc.ready) return fail(c, Q3C_ERR_STATE, "no finalized encoder in this context");
    if (!audio || !codes || !ws || B < 1 || samples < 1) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_encode: bad argument");
    if ((uintptr_t)ws & 255) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
    size_t need = 0;
    TRY(encode_sequence(c, B, samples, false, nullptr, nullptr, nullptr, nullptr, nullptr, &need));
    if (ws_bytes < need) return fail(c, Q3C_ERR_WORKSPACE, "workspace too small: need " + std::to_string(need));
    return encode_sequence(c, B, samples, true, audio, codes, embeddings, (char*)ws, (hipStream_t)stream, nullptr);
}

}  // extern "C"
