// Host side of the q3c C ABI: context, weight repacking, and the launch sequence of one decode forward.
// Declarations and their reference citations are in include/q3c.h.
#include "../../include/q3c.h"
#include "common.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#ifdef Q3C_HOST_DRYRUN
// Sanitizer build of the HOST side only (tools/sanitize_host.py): "device" memory is host memory, so q3c_finalize's weight
// repacking, the workspace / stream-state layouts and every argument check run under AddressSanitizer and UBSan on a machine
// without a GPU.  Nothing is launched in such a build and it is never what the package loads.
#include <cstdlib>
static hipError_t dry_malloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
alignas(256) static unsigned char dry_static_scratch[4096];
#define hipMalloc(p, n) dry_malloc((void**)(p), (n))
#define hipFree(p) (std::free(p), hipSuccess)
#define hipMemcpy(d, s, n, kind) (std::memcpy((d), (s), (n)), hipSuccess)
#define hipMemset(d, v, n) (std::memset((d), (v), (n)), hipSuccess)
#define hipGetDevice(p) (*(p) = 0, hipSuccess)
#define hipGetSymbolAddress(pp, sym) (*(pp) = (void*)dry_static_scratch, hipSuccess)
#endif

// Workspace of q3c_forward_static: one array per GPU the library's code object is loaded on, held until the process ends.
// Host code never uses this name as an address: q3c_finalize asks the runtime for the array's address on the context's
// device and everything after that goes through the pointer it returned.
static __device__ __align__(256) unsigned char q3c_static_scratch[Q3C_STATIC_SCRATCH_BYTES];

namespace {

struct HostTensor { std::vector<int64_t> shape; std::vector<float> data; };

struct GemmW {            // one packed contraction layer
    void* w = nullptr;    // [taps][N][K], element type in_dtype
    float* bias = nullptr;
    int taps = 1, N = 0, K = 0, off0 = 0, off_step = 0, in_dtype = Q3_F32;
};
struct Snake { float* a = nullptr; float* invb = nullptr; int C = 0; };

struct Layer { float* ln1; float* ln2; GemmW qkv, o, gu, down; float* ls_attn; float* ls_mlp; };
struct UpStage { GemmW tconv; int f; float *dw_w, *dw_b, *ln_w, *ln_b; GemmW pw1, pw2; float* gamma; };
struct Unit { Snake a1, a2; GemmW c1, c2; };
struct Block {
    Snake head; GemmW tconv; int rate; std::vector<Unit> units;
    Snake first_act_tiled;     // units[0].a1 repeated `rate` times: the transposed conv's columns are (phase, channel)
};

// the encode side (encoder_impl.h)
struct EncStage { GemmW c3, c1, down; int r = 1, dim = 0; };
struct EncLayer { float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *ls_attn, *ls_mlp; GemmW qkv, o, fc1, fc2; };
struct EncoderState {
    q3c_encoder_config cfg{};
    bool configured = false, ready = false;
    float *conv0_w = nullptr, *conv0_b = nullptr;     // first conv, fp32 [F][k] and [F]
    std::vector<EncStage> stages;
    GemmW last, ds;
    std::vector<EncLayer> layers;
    float *cos_t = nullptr, *sin_t = nullptr;
};

uint16_t f32_to_bf16(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

}  // namespace

struct q3c_ctx {
    q3c_config cfg{};
    int dtype = Q3_F32;
    mutable std::string err;
    std::map<std::string, HostTensor> staged;
    bool finalized = false;
    std::vector<void*> allocs;

    // front end
    int G = 0;
    std::vector<float*> tables;
    float** tables_dev = nullptr;
    int* group_start_dev = nullptr;
    float* emb_table = nullptr;
    void** tables_t_dev = nullptr;    // per code book a dimension-major copy [D/4][bins][4] followed by the rows' squared norms [bins]
    float* rvq_in_proj = nullptr;     // [G][codebook_dim][rvq_in_dim]: the groups' input_proj (encode direction), when loaded
    int rvq_in_dim = 0;
    GemmW rvq_proj, pre_conv, in_proj, out_proj;
    // transformer
    std::vector<Layer> layers;
    float* final_norm = nullptr;
    float *cos_t = nullptr, *sin_t = nullptr;
    // conv decoder
    std::vector<UpStage> ups;
    GemmW conv_in;
    std::vector<Block> blocks;
    Snake final_snake;
    GemmW conv_out;
    void* conv_out_fold = nullptr;    // bf16 contexts: conv_out's taps as [16][channels] (rows past its 7 taps zero), the operand of the
                                      // last residual unit's folded product (ConvGemmArgs::W3)

    int device = -1;                  // the device that was current in q3c_finalize
    void* static_scratch = nullptr;   // address of q3c_static_scratch on that device
    int* err_flag = nullptr;
    void* zero_page = nullptr;        // 256 zero bytes: what the kernels read for rows outside a sequence
    std::map<int, std::pair<float*, size_t>> taps;

    // per-kernel timing
    struct ProfRec { hipEvent_t a, b; const char* name; double flops, bytes; };
    bool fuse_units = true;           // residual units of <= 192 channels as one launch
    int tuning = 0;                   // measurement switches of THIS context (q3c_debug_set_tuning; 0 = defaults): tile-shape choices
                                      // (ConvGemmArgs::tuning), bit 256 the vector-ALU attention, bit 512 the final conv as its own
                                      // launch, bit 1 << 20 the vector-ALU code-book search
    long long* timeline = nullptr;    // measurement aid (q3c_debug_set_timeline): phase timestamps of every 64th workgroup
    int timeline_slots = 0;
    bool profiling = false, profile_detail = false;
    std::set<std::string> interned;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> event_pool;

    int total_upsample = 1, trim = 0;

    EncoderState enc;                 // the tokenizer's encode side, when q3c_encoder_configure was called
};

namespace {

int fail(const q3c_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }
int encoder_finalize(q3c_ctx* c);     // encoder_impl.h

// Brackets one launch with events when profiling is on.
struct ProfScope {
    q3c_ctx* c; hipStream_t s; q3c_ctx::ProfRec r{}; bool on;
    ProfScope(q3c_ctx* c_, hipStream_t s_, const char* name, double flops, double bytes) : c(c_), s(s_), on(c_->profiling) {
        if (!on) return;
        auto get = [&]() { hipEvent_t e; if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
        r.a = get(); r.b = get(); r.name = name; r.flops = flops; r.bytes = bytes;
        (void)hipEventRecord(r.a, s);
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(r.b, s); c->prof.push_back(r); } }
};

#define HIPCHK(ctx, expr)                                                                              \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(ctx, Q3C_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
    } while (0)

size_t esize(int dtype) { return dtype == Q3_BF16 ? 2 : 4; }
size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// ---- uploads ------------------------------------------------------------------------------------------
int dev_alloc(q3c_ctx* c, size_t bytes, void** out) {
    void* p = nullptr;
    HIPCHK(c, hipMalloc(&p, bytes ? bytes : 16));
    c->allocs.push_back(p);
    *out = p;
    return 0;
}
int upload_f32(q3c_ctx* c, const std::vector<float>& v, float** out) {
    void* p; int r = dev_alloc(c, v.size() * 4, &p); if (r) return r;
    HIPCHK(c, hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    *out = (float*)p;
    return 0;
}
int upload_typed(q3c_ctx* c, const std::vector<float>& v, int dtype, void** out) {
    if (dtype == Q3_F32) { float* p; int r = upload_f32(c, v, &p); *out = p; return r; }
    std::vector<uint16_t> h(v.size());
    for (size_t i = 0; i < v.size(); ++i) h[i] = f32_to_bf16(v[i]);
    void* p; int r = dev_alloc(c, h.size() * 2, &p); if (r) return r;
    HIPCHK(c, hipMemcpy(p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    *out = p;
    return 0;
}

// ---- weight lookup ------------------------------------------------------------------------------------
const HostTensor* find(q3c_ctx* c, const std::string& name, std::initializer_list<int64_t> shape, int* rc) {
    auto it = c->staged.find(name);
    if (it == c->staged.end()) { *rc = fail(c, Q3C_ERR_WEIGHT, "missing weight " + name); return nullptr; }
    std::vector<int64_t> want(shape);
    if (it->second.shape != want) {
        std::string s = "weight " + name + " has shape [";
        for (auto d : it->second.shape) s += std::to_string(d) + ",";
        s += "], expected [";
        for (auto d : want) s += std::to_string(d) + ",";
        *rc = fail(c, Q3C_ERR_WEIGHT, s + "]");
        return nullptr;
    }
    return &it->second;
}
#define GETW(var, name, ...)                                   \
    const HostTensor* var = find(c, name, {__VA_ARGS__}, &rc); \
    if (!var) return rc

// causal Conv1d weight [Cout][Cin][k] -> [k][Cout][Cin]
int pack_conv(q3c_ctx* c, const std::string& prefix, int cout, int cin, int k, int dil, int dtype, GemmW* g, bool has_bias = true) {
    int rc = 0;
    GETW(w, prefix + ".weight", cout, cin, k);
    std::vector<float> p((size_t)k * cout * cin);
    for (int n = 0; n < cout; ++n)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < k; ++t) p[((size_t)t * cout + n) * cin + ci] = w->data[((size_t)n * cin + ci) * k + t];
    g->taps = k; g->N = cout; g->K = cin; g->off0 = -(k - 1) * dil; g->off_step = dil; g->in_dtype = dtype;
    if ((rc = upload_typed(c, p, dtype, &g->w))) return rc;
    if (has_bias) {
        GETW(b, prefix + ".bias", cout);
        if ((rc = upload_f32(c, b->data, &g->bias))) return rc;
    }
    return 0;
}
// Linear weight [out][in] (+ optional bias)
int pack_linear(q3c_ctx* c, const std::string& wname, const std::string& bname, int out, int in, int dtype, GemmW* g) {
    int rc = 0;
    GETW(w, wname, out, in);
    g->taps = 1; g->N = out; g->K = in; g->off0 = 0; g->off_step = 0; g->in_dtype = dtype;
    if ((rc = upload_typed(c, w->data, dtype, &g->w))) return rc;
    if (!bname.empty()) {
        GETW(b, bname, out);
        if ((rc = upload_f32(c, b->data, &g->bias))) return rc;
    }
    return 0;
}
// ConvTranspose1d weight [Cin][Cout][k], stride s, k == s or k == 2s, as a GEMM over N = s*Cout columns:
//   k == 2s: y[q*s+p] = x[q] . w[:, :, p+s] + x[q+1] . w[:, :, p]      (after the two-sided trim of k - s)
//   k ==  s: y[q*s+p] = x[q] . w[:, :, p]
int pack_tconv_host(q3c_ctx* c, const float* w, const float* b, int cin, int cout, int k, int s, int dtype, GemmW* g) {
    if (k != s && k != 2 * s) return fail(c, Q3C_ERR_UNSUPPORTED, "transposed conv needs kernel == stride or 2*stride");
    const int taps = k / s, N = s * cout;
    std::vector<float> p((size_t)taps * N * cin);
    for (int t = 0; t < taps; ++t)
        for (int ph = 0; ph < s; ++ph)
            for (int co = 0; co < cout; ++co)
                for (int ci = 0; ci < cin; ++ci) {
                    int kk = (taps == 2) ? (t == 0 ? ph + s : ph) : ph;
                    p[((size_t)t * N + ph * cout + co) * cin + ci] = w[((size_t)ci * cout + co) * k + kk];
                }
    g->taps = taps; g->N = N; g->K = cin; g->off0 = 0; g->off_step = 1; g->in_dtype = dtype;
    int rc;
    if ((rc = upload_typed(c, p, dtype, &g->w))) return rc;
    if (b) {
        std::vector<float> bb((size_t)N);
        for (int ph = 0; ph < s; ++ph)
            for (int co = 0; co < cout; ++co) bb[(size_t)ph * cout + co] = b[co];
        if ((rc = upload_f32(c, bb, &g->bias))) return rc;
    }
    return 0;
}
int pack_tconv(q3c_ctx* c, const std::string& prefix, int cin, int cout, int k, int s, int dtype, GemmW* g) {
    int rc = 0;
    GETW(w, prefix + ".weight", cin, cout, k);
    GETW(b, prefix + ".bias", cout);
    return pack_tconv_host(c, w->data.data(), b->data.data(), cin, cout, k, s, dtype, g);
}
int pack_snake_host(q3c_ctx* c, const float* alpha, const float* beta, int C, float eps, Snake* s) {
    std::vector<float> va(C), vb(C);
    // bf16 contexts evaluate the sine with v_sin_f32, whose argument is in revolutions: fold 1/(2 pi) into the frequency
    const float fold = c->dtype == Q3_BF16 ? 0.15915494309189535f : 1.0f;
    for (int i = 0; i < C; ++i) { va[i] = expf(alpha[i]) * fold; vb[i] = 1.0f / (expf(beta[i]) + eps); }
    s->C = C;
    int rc;
    if ((rc = upload_f32(c, va, &s->a))) return rc;
    return upload_f32(c, vb, &s->invb);
}
int pack_snake(q3c_ctx* c, const std::string& prefix, int C, Snake* s, int repeat = 1) {
    int rc = 0;
    GETW(a, prefix + "alpha", C);
    GETW(b, prefix + "beta", C);
    if (repeat == 1) return pack_snake_host(c, a->data.data(), b->data.data(), C, c->cfg.snake_eps, s);
    std::vector<float> ta, tb;
    for (int r = 0; r < repeat; ++r) { ta.insert(ta.end(), a->data.begin(), a->data.end()); tb.insert(tb.end(), b->data.begin(), b->data.end()); }
    return pack_snake_host(c, ta.data(), tb.data(), C * repeat, c->cfg.snake_eps, s);
}
int pack_vec(q3c_ctx* c, const std::string& name, int n, float** out) {
    int rc = 0;
    GETW(v, name, n);
    return upload_f32(c, v->data, out);
}

int ensure_zero_page(q3c_ctx* c) {
    if (c->zero_page) return 0;
    int rc = dev_alloc(c, 256, &c->zero_page);
    if (rc) return rc;
    HIPCHK(c, hipMemset(c->zero_page, 0, 256));
    return 0;
}

// ---- one convgemm launch --------------------------------------------------------------------------------
struct GemmOpts {
    int act = Q3_ACT_NONE;
    const Snake* epi = nullptr;
    const float* scale = nullptr;
    const void* R = nullptr;
    const Snake* post = nullptr;
    bool post_elu = false;            // the post activation is ELU (the encode side), not a SnakeBeta
    void* Y2 = nullptr;               // second output: Y gets the pre-`post` value, Y2 the post value
    const GemmW* second = nullptr;    // fused 1x1 second contraction (residual unit in one launch)
    const Snake* pre = nullptr;       // with `second`: SnakeBeta applied to the staged input tile (the unit reads the raw stream)
    const void* fold_w = nullptr;     // with `second` and `post`: the output is not stored; its products with the 16 rows of fold_w
    float* fold_p = nullptr;          // go to fold_p, [B][M][8] fp32 (the following one-channel conv, finished by q3_launch_tap_sum)
    // Streaming layouts.  `segmented`: the B items' rows are stacked in one row space, each item = x_hist carried rows
    // followed by its L_in new rows, and only new rows produce output.  Offsets / strides are in elements of the
    // respective tensor; a stride of 0 means dense ([B][M][n_out]).
    bool segmented = false;
    int x_hist = 0;
    int off_shift = 0;                // added to the layer's first tap offset
    long long y_off = 0, y_bstride = 0, y2_off = 0, y2_bstride = 0, r_off = 0, r_bstride = 0;
    long long x_bstride = 0;          // dense layout only: the input's items are this many elements apart (0: L_in rows)
    const int* row_start = nullptr;   // [B]: output rows below it are stored as zeros
    int row0 = 0;                     // dense layout only: output rows below row0 are NOT computed (nobody reads them); their
                                      // inputs exist, so the rows in front of row0 serve as the halo
};

// X [B][L_in][K] contiguous -> Y [B][M][N_out] contiguous (N_out = N, or N/2 for SwiGLU); R like Y.
int gemm(q3c_ctx* c, const GemmW& g, const void* X, int B, int L_in, void* Y, int M, int out_dtype, const GemmOpts& o,
         hipStream_t s) {
    ConvGemmArgs a{};
    const int n_out = o.act == Q3_ACT_SWIGLU ? g.N / 2 : g.N;
    a.X = X; a.ldx = g.K; a.x_batch_stride = (long long)L_in * g.K; a.L_in = L_in;
    a.W = g.w; a.bias = g.bias;
    a.Y = Y; a.ldy = n_out; a.y_batch_stride = (long long)M * n_out;
    a.R = o.R; a.ldr = n_out; a.r_batch_stride = (long long)M * n_out;
    a.B = B; a.M = M; a.N = g.N; a.K = g.K;
    a.taps = g.taps; a.off0 = g.off0; a.off_step = g.off_step;
    a.act = o.act;
    if (o.epi) { a.epi_a = o.epi->a; a.epi_invb = o.epi->invb; }
    a.scale = o.scale;
    if (o.post) { a.post_a = o.post->a; a.post_invb = o.post->invb; }
    a.post_elu = o.post_elu ? 1 : 0;
    a.Y2 = o.Y2; a.y2_batch_stride = a.y_batch_stride;
    a.zeros = c->zero_page;
    a.row_start = o.row_start;
    a.tuning = c->tuning; a.timeline = c->timeline; a.timeline_slots = c->timeline_slots;
    if (o.second) {
        if (!o.pre) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "fused residual unit without its input SnakeBeta");
        a.W2 = o.second->w; a.bias2 = o.second->bias; a.pre_a = o.pre->a; a.pre_invb = o.pre->invb;
        if (o.fold_w) { a.W3 = o.fold_w; a.P = o.fold_p; a.p_batch_stride = (long long)M * 8; }
    }
    if (o.segmented) {
        if (M != L_in) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "segmented contraction needs as many output as input rows");
        const size_t eo = esize(out_dtype);
        a.seg_len = o.x_hist + L_in; a.seg_hist = o.x_hist;
        a.B = 1; a.M = B * a.seg_len; a.L_in = a.M; a.x_batch_stride = 0;
        a.off0 = g.off0 + o.off_shift;
        a.Y = (char*)Y + o.y_off * eo;
        if (o.y_bstride) a.y_batch_stride = o.y_bstride;
        if (o.Y2) { a.Y2 = (char*)o.Y2 + o.y2_off * eo; a.y2_batch_stride = o.y2_bstride ? o.y2_bstride : a.y2_batch_stride; }
        if (o.R) { a.R = (const char*)o.R + o.r_off * eo; if (o.r_bstride) a.r_batch_stride = o.r_bstride; }
    } else {
        // dense inputs whose outputs go to item-strided tensors (strides in elements of the output type)
        if (o.x_bstride) a.x_batch_stride = o.x_bstride;
        if (o.y_bstride) a.y_batch_stride = o.y_bstride;
        if (o.Y2 && o.y2_bstride) a.y2_batch_stride = o.y2_bstride;
        if (g.taps == 1 && M == L_in && !o.x_bstride && !o.y_bstride && !(o.Y2 && o.y2_bstride) && !o.row0) {      // no halo: batch items are just more rows
            a.M = B * M; a.L_in = B * L_in; a.B = 1;
        }
        if (o.row0 > 0) {             // every tensor starts row0 rows further on; the item strides stay those of the whole tensors
            if (o.row0 >= M) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "first computed row beyond the output");
            const size_t ei = esize(g.in_dtype), eo = esize(out_dtype);
            a.X = (const char*)a.X + (size_t)o.row0 * a.ldx * ei;
            a.Y = (char*)a.Y + (size_t)o.row0 * a.ldy * eo;
            if (a.Y2) a.Y2 = (char*)a.Y2 + (size_t)o.row0 * a.ldy * eo;
            if (a.R) a.R = (const char*)a.R + (size_t)o.row0 * a.ldr * eo;
            if (a.P) a.P += (size_t)o.row0 * 8;
            a.shape_rows = a.M;       // same tile shape, hence the same summation order, as the whole launch
            a.M -= o.row0; a.L_in -= o.row0; a.row_lo = -o.row0;
        }
    }
    const int Mc = o.segmented ? M : M - o.row0;      // rows computed per item
    const char* pname = "";
    if (c->profiling) {
        pname = q3_convgemm_variant(a, g.in_dtype, out_dtype);
        if (c->profile_detail) {      // one entry per layer shape: interned so the pointer outlives this call
            char buf[64];
            std::snprintf(buf, sizeof(buf), "%.24s M%dxN%dxK%dxT%d%s", pname + 9, B * Mc, a.N, a.K, a.taps, o.second ? "+1x1" : "");
            pname = c->interned.insert(buf).first->c_str();
        }
    }
    // algorithmic bytes: input, weights, and each output / residual tensor once
    // (the fused unit's residual rows are rows of its input tensor, read moments before: not a second stream)
    const double abytes = (double)B * (L_in - (o.segmented ? 0 : o.row0)) * g.K * esize(g.in_dtype) + (double)g.taps * g.N * g.K * esize(g.in_dtype) +
                          (double)B * Mc * n_out * esize(out_dtype) * (1 + (o.Y2 ? 1 : 0) + ((o.R && !o.second) ? 1 : 0));
    ProfScope ps(c, s, pname, 2.0 * B * Mc * (double)a.N * (a.K * a.taps + (o.second ? a.N : 0)), abytes);
    hipError_t e = q3_launch_convgemm(a, g.in_dtype, out_dtype, s);
    if (e != hipSuccess) return fail(c, Q3C_ERR_HIP, std::string("convgemm launch: ") + hipGetErrorString(e));
    return 0;
}

// ---- workspace layout -----------------------------------------------------------------------------------
struct Layout {
    size_t S = 0, Qb = 0, A = 0, Bq = 0, C = 0, D = 0;    // region sizes in bytes
    size_t oS = 0, oQ = 0, oA = 0, oB = 0, oC = 0, oD = 0, total = 0;
};

// Walks the stage sequence of forward() and records the largest tenant of every region.
Layout make_layout(const q3c_ctx* c, int B, int T) {
    const q3c_config& k = c->cfg;
    const size_t es = esize(c->dtype);
    const size_t fr = (size_t)B * T;
    Layout l;
    auto mx = [](size_t& a, size_t b) { if (b > a) a = b; };
    if (k.frontend == Q3C_FRONTEND_SPLIT_RVQ) {
        l.S = fr * c->G * k.codebook_dim * 4;
        l.Qb = fr * k.rvq_out_dim * es;
    }
    const size_t wmax = (size_t)std::max(k.latent_dim, k.hidden_size);
    mx(l.A, fr * wmax * es); mx(l.Bq, fr * wmax * es);
    const size_t qd = (size_t)k.num_attention_heads * k.head_dim, kvd = (size_t)k.num_key_value_heads * k.head_dim;
    mx(l.C, align_up(fr * (qd + 2 * kvd) * es) + align_up(fr * qd * es) + align_up(fr * k.intermediate_size * es));
    size_t L = T;
    for (int i = 0; i < k.num_upsampling_ratios; ++i) {
        L *= k.upsampling_ratios[i];
        size_t rows = (size_t)B * L;
        mx(l.A, rows * k.latent_dim * es); mx(l.Bq, rows * k.latent_dim * es);
        mx(l.C, align_up(rows * k.latent_dim * es) + rows * 4 * k.latent_dim * es);
    }
    mx(l.A, (size_t)B * L * k.decoder_dim * es); mx(l.Bq, (size_t)B * L * k.decoder_dim * es);
    int ch = k.decoder_dim;
    for (int i = 0; i < k.num_upsample_rates; ++i) {
        L = (L - 1) * k.upsample_rates[i];
        ch /= 2;
        size_t bytes = (size_t)B * L * ch * es;
        // a block keeps four tensors alive: its input, the raw stream, and two snaked copies that alternate
        mx(l.A, bytes); mx(l.Bq, bytes); mx(l.C, bytes); mx(l.D, bytes);
    }
    l.S = align_up(l.S); l.Qb = align_up(l.Qb); l.A = align_up(l.A); l.Bq = align_up(l.Bq); l.C = align_up(l.C); l.D = align_up(l.D);
    l.oS = 0; l.oQ = l.oS + l.S; l.oA = l.oQ + l.Qb; l.oB = l.oA + l.A; l.oC = l.oB + l.Bq; l.oD = l.oC + l.C;
    l.total = l.oD + l.D;
    return l;
}

int tap_out(q3c_ctx* c, int stage, const void* p, size_t n, hipStream_t s) {
    auto it = c->taps.find(stage);
    if (it == c->taps.end() || !it->second.first) return 0;
    if (n > it->second.second) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "tap buffer too small for stage " + std::to_string(stage));
    hipError_t e = q3_launch_cast_to_f32(p, it->second.first, (long long)n, c->dtype, s);
    if (e != hipSuccess) return fail(c, Q3C_ERR_HIP, std::string("tap cast: ") + hipGetErrorString(e));
    return 0;
}

#define LAUNCH(ctx, expr)                                                                            \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) return fail(ctx, Q3C_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
// a memory-bound launch with its algorithmic byte count
#define LAUNCH_B(ctx, stream, name, bytes, expr)                 \
    do {                                                         \
        ProfScope _ps(ctx, stream, name, 0.0, (double)(bytes));  \
        LAUNCH(ctx, expr);                                       \
    } while (0)
#define TRY(expr) do { int _r = (expr); if (_r) return _r; } while (0)

// in_proj -> sliding-window layers -> final RMSNorm -> out_proj on x0 [B,T,latent] held in bufA; bufB is as large, bufC
// holds q/k/v, the attention output and the MLP activations.  *cur is the result, *other the free one of bufA / bufB.
int transformer_stack(q3c_ctx* c, int B, int T, void* bufA, void* bufB, char* bufC, hipStream_t s, void** cur_out, void** other_out) {
    const q3c_config& k = c->cfg;
    const int dt = c->dtype;
    const size_t es = esize(dt);
    const size_t fr = (size_t)B * T;
    const int H = k.hidden_size;
    void* x = bufA; void* nrm = bufB;
    if (k.io_proj) { TRY(gemm(c, c->in_proj, bufA, B, T, bufB, T, dt, GemmOpts{}, s)); x = bufB; nrm = bufA; }
    const int qd = k.num_attention_heads * k.head_dim, kvd = k.num_key_value_heads * k.head_dim;
    void* qkv = bufC;
    void* att = bufC + align_up(fr * (qd + 2 * kvd) * es);
    void* mlp = (char*)att + align_up(fr * qd * es);
    for (int l = 0; l < k.num_hidden_layers; ++l) {
        const Layer& y = c->layers[l];
        LAUNCH_B(c, s, "rmsnorm", 2 * fr * H * es, q3_launch_rmsnorm(x, y.ln1, nrm, (long long)fr, H, k.rms_norm_eps, dt, s));
        TRY(gemm(c, y.qkv, nrm, B, T, qkv, T, dt, GemmOpts{}, s));
        LAUNCH_B(c, s, "swa_attention", fr * (size_t)(2 * qd + 2 * kvd) * es,
                 q3_launch_swa_attention(qkv, att, c->cos_t, c->sin_t, B, T, k.num_attention_heads, k.num_key_value_heads, k.head_dim,
                                         k.sliding_window, dt, s, 0, nullptr, 1 << 30, nullptr, c->tuning));
        GemmOpts o1; o1.scale = y.ls_attn; o1.R = x;
        TRY(gemm(c, y.o, att, B, T, x, T, dt, o1, s));
        LAUNCH_B(c, s, "rmsnorm", 2 * fr * H * es, q3_launch_rmsnorm(x, y.ln2, nrm, (long long)fr, H, k.rms_norm_eps, dt, s));
        GemmOpts o2; o2.act = Q3_ACT_SWIGLU;
        TRY(gemm(c, y.gu, nrm, B, T, mlp, T, dt, o2, s));
        GemmOpts o3; o3.scale = y.ls_mlp; o3.R = x;
        TRY(gemm(c, y.down, mlp, B, T, x, T, dt, o3, s));
    }
    LAUNCH_B(c, s, "rmsnorm", 2 * fr * H * es, q3_launch_rmsnorm(x, c->final_norm, nrm, (long long)fr, H, k.rms_norm_eps, dt, s));
    if (k.io_proj) { TRY(gemm(c, c->out_proj, nrm, B, T, x, T, dt, GemmOpts{}, s)); *cur_out = x; *other_out = nrm; }
    else { *cur_out = nrm; *other_out = x; }
    return 0;
}

int validate_config(q3c_ctx* c) {
    const q3c_config& k = c->cfg;
    auto bad = [&](const char* m) { return fail(c, Q3C_ERR_INVALID_ARGUMENT, m); };
    if (k.struct_size != (int)sizeof(q3c_config)) return bad("q3c_config.struct_size mismatch");
    if (k.precision != Q3C_PRECISION_F32 && k.precision != Q3C_PRECISION_BF16) return bad("unknown precision");
    if (k.frontend != Q3C_FRONTEND_SPLIT_RVQ && k.frontend != Q3C_FRONTEND_MEAN_EMBEDDING) return bad("unknown frontend");
    if (k.num_upsampling_ratios < 0 || k.num_upsampling_ratios > Q3C_MAX_STAGES || k.num_upsample_rates < 1 ||
        k.num_upsample_rates > Q3C_MAX_STAGES || k.num_dilations < 1 || k.num_dilations > Q3C_MAX_STAGES)
        return bad("stage counts out of range");
    if (k.codebook_size < 1 || k.num_quantizers < 1 || k.num_semantic_quantizers < 1 || k.num_semantic_quantizers > k.num_quantizers)
        return bad("codebook sizes out of range");
    if (k.num_attention_heads < 1 || k.num_key_value_heads < 1 || k.head_dim < 2 || k.hidden_size < 1 || k.latent_dim < 1 ||
        k.intermediate_size < 1 || k.num_hidden_layers < 0 || k.decoder_dim < 1)
        return bad("sizes must be positive");
    if (k.frontend == Q3C_FRONTEND_SPLIT_RVQ && (k.codebook_dim < 1 || k.rvq_out_dim < 1 || k.pre_conv_kernel < 1))
        return bad("front-end sizes must be positive");
    for (int i = 0; i < k.num_upsampling_ratios; ++i) if (k.upsampling_ratios[i] < 1) return bad("upsampling_ratios must be >= 1");
    for (int i = 0; i < k.num_upsample_rates; ++i) if (k.upsample_rates[i] < 1) return bad("upsample_rates must be >= 1");
    for (int i = 0; i < k.num_dilations; ++i) if (k.dilations[i] < 1) return bad("dilations must be >= 1");
    if (k.num_attention_heads % k.num_key_value_heads || (k.head_dim & 1)) return bad("attention head layout");
    if (k.frontend == Q3C_FRONTEND_MEAN_EMBEDDING && (k.io_proj || k.latent_dim != k.hidden_size))
        return bad("mean_embedding front end needs io_proj = 0 and latent_dim == hidden_size");
    if (!k.io_proj && k.latent_dim != k.hidden_size) return bad("latent_dim != hidden_size needs io_proj");
    if (k.max_positions < 1) return bad("max_positions");
    // every contraction width must be a whole number of 16-byte pieces
    int widths[] = {k.latent_dim, k.hidden_size, k.num_attention_heads * k.head_dim, k.num_key_value_heads * k.head_dim,
                    k.intermediate_size, k.decoder_dim >> k.num_upsample_rates};
    for (int w : widths) if (w % 8) return fail(c, Q3C_ERR_UNSUPPORTED, "channel counts must be multiples of 8");
    if (k.frontend == Q3C_FRONTEND_SPLIT_RVQ && (k.codebook_dim % 4 || k.rvq_out_dim % 8))
        return fail(c, Q3C_ERR_UNSUPPORTED, "codebook_dim must be a multiple of 4 and rvq_out_dim of 8");
    if (k.intermediate_size % 32) return fail(c, Q3C_ERR_UNSUPPORTED, "intermediate_size must be a multiple of 32");
    if (k.decoder_dim % (1 << k.num_upsample_rates)) return bad("decoder_dim not divisible by 2^blocks");
    if (k.sliding_window < 1 || k.sliding_window > 253) return fail(c, Q3C_ERR_UNSUPPORTED, "sliding_window must be in [1, 253]");
    return 0;
}


// =========================================================================================================
// Streaming decode.  Every layer of the decode is causal, so a stream can be advanced a few frames at a time and
// produce exactly the samples of the un-chunked forward, provided each look-back (conv taps, the previous input row
// of a transposed conv, the attention window) finds the rows of the previous pushes.  Those carried rows are the
// per-slot state record; a push over S slots stacks [carried | new] rows of all S streams in one row space per
// layer, so one launch per layer serves every stream whatever its position.
//
// A transposed conv with kernel 2r, stride r emits, for new input rows x[a..a+n), the r-row groups that need
// (x[j-1], x[j]); the group of (nothing, x[0]) is what the reference trims on the left, so on a stream's first push
// the leading rows of every later stage do not exist: they are stored as zeros (exactly the zero padding the causal
// convs of the un-chunked forward see there) and the first `trim` samples of the first push are blank.
// =========================================================================================================
// a captured push: the launch sequence of one (S, n) with these exact operand addresses baked into its kernel nodes
struct PushGraph { const void *slots, *codes, *wav, *ws; int tuning; bool fuse; hipGraphExec_t exec; unsigned long long used; };
struct StreamPlan {
    size_t ws_bytes = 0; Q3CarryDesc* descs = nullptr; int ndesc = 0; std::vector<PushGraph> graphs;
    double carried_bytes = 0, appended_bytes = 0;      // per stream: rows copied in and out around a push; rows appended to rings
};
constexpr size_t MAX_GRAPHS_PER_PLAN = 32;

}  // namespace

struct q3c_stream_pool {
    q3c_ctx* c = nullptr;
    int capacity = 0;
    char* state = nullptr;
    long long slot_stride = 0;
    int* pos = nullptr;
    std::map<std::pair<int, int>, StreamPlan> plans;
    std::vector<void*> allocs;
    hipStream_t capture_stream = nullptr;   // side stream the push graphs are captured on (the legacy default stream cannot capture)
    unsigned long long graph_clock = 0;     // replay counter: least-recently-used eviction of cached graphs
};

namespace {

struct SBuf { size_t off; int hist, rows, ch; size_t es; };   // per item: (hist + rows) rows of ch elements

// The launch sequence of one push; with run == false it only lays out the workspace and lists the carried tensors.
int stream_sequence(q3c_stream_pool* P, int S, int n, bool run, const StreamPlan* plan, const int* slots, const int64_t* codes,
                    float* wav, char* ws, hipStream_t s, std::vector<Q3CarryDesc>* descs, size_t* ws_total, long long* state_total) {
    q3c_ctx* c = P->c;
    const q3c_config& k = c->cfg;
    const int dt = c->dtype;
    const size_t es = esize(dt);
    size_t off = 0;
    long long st_off = 0;
    // the layout pass (run == false) has no workspace: offsets are added as integers, never to a null pointer
    const uintptr_t ws_base = (uintptr_t)ws;
    auto W = [&](size_t o) { return (char*)(ws_base + o); };
    auto alloc = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    auto carried = [&](const SBuf& b, int col_off_elems, int cols) {
        Q3CarryDesc d{};
        d.work_off = (long long)b.off; d.work_bstride = (long long)(b.hist + b.rows) * b.ch * b.es;
        d.state_off = st_off; d.work_ld = (int)(b.ch * b.es); d.col_off = (int)(col_off_elems * b.es); d.row_bytes = (int)(cols * b.es);
        d.rows = b.hist; d.new_rows = b.rows;
        if (descs) descs->push_back(d);
        st_off += (long long)align_up((size_t)d.rows * d.row_bytes, 16);
    };
    auto tensor = [&](int hist, int rows, int ch, size_t e, int carry_col = 0, int carry_cols = -1) {
        SBuf b{alloc((size_t)S * (hist + rows) * ch * e), hist, rows, ch, e};
        if (hist) carried(b, carry_col, carry_cols < 0 ? ch : carry_cols);
        return b;
    };
    auto ptr = [&](const SBuf& b) { return (void*)W(b.off); };
    // output / residual placement of a tensor with carried rows, in elements
    auto new_rows_off = [](const SBuf& b) { return (long long)b.hist * b.ch; };
    auto item_stride = [](const SBuf& b) { return (long long)(b.hist + b.rows) * b.ch; };

    const int nb = k.num_upsample_rates;
    const size_t o_pos0 = alloc((size_t)S * 4);
    const size_t o_rs = alloc((size_t)S * 4 * 2 * nb);
    int* pos0 = (int*)W(o_pos0);
    int* rs_all = (int*)W(o_rs);
    Q3StageStarts st{};
    st.count = 2 * nb;
    {
        int d_in = 0;
        for (int i = 0; i < nb; ++i) { st.rows[2 * i] = d_in + 1; d_in = (d_in + 1) * k.upsample_rates[i]; st.rows[2 * i + 1] = d_in; }
    }
    if (run) {
        LAUNCH(c, q3_launch_stream_prep(slots, S, P->capacity, P->pos, n, k.max_positions, pos0, rs_all, st, c->err_flag, s));
        LAUNCH_B(c, s, "stream_carry", 2.0 * S * plan->carried_bytes,
                 q3_launch_stream_carry(plan->descs, plan->ndesc, ws, P->state, P->slot_stride, slots, S, P->capacity, 0, pos0, s));
    }

    const int H = k.hidden_size, Lt = k.latent_dim;
    const size_t fr = (size_t)S * n;
    const size_t wmax = (size_t)std::max(Lt, H);
    const size_t o_x = alloc(fr * wmax * es), o_nrm = alloc(fr * wmax * es);
    void* bufA = W(o_x); void* bufB = W(o_nrm);

    // ---- front end ----
    if (k.frontend == Q3C_FRONTEND_MEAN_EMBEDDING) {
        if (run) LAUNCH_B(c, s, "mean_embedding", fr * ((size_t)k.num_quantizers * (8 + H * 4) + H * es),
                          q3_launch_mean_embedding((const long long*)codes, S, k.num_quantizers, n, k.codebook_size, H, c->emb_table, bufA,
                                                   dt, c->err_flag, s));
    } else {
        const size_t o_sum = alloc(fr * c->G * k.codebook_dim * 4);
        SBuf qb = tensor(k.pre_conv_kernel - 1, n, k.rvq_out_dim, es);
        if (run) {
            LAUNCH_B(c, s, "rvq_gather_sum", fr * ((size_t)k.num_quantizers * (8 + k.codebook_dim * 4) + (size_t)c->G * k.codebook_dim * 4),
                     q3_launch_rvq_gather_sum((const long long*)codes, S, k.num_quantizers, n, k.codebook_size, k.codebook_dim,
                                              c->tables_dev, c->group_start_dev, c->G, (float*)W(o_sum), c->err_flag, s));
            GemmOpts g; g.segmented = true; g.y_off = new_rows_off(qb); g.y_bstride = item_stride(qb);
            TRY(gemm(c, c->rvq_proj, W(o_sum), S, n, ptr(qb), n, dt, g, s));
            GemmOpts g2; g2.segmented = true; g2.x_hist = qb.hist;
            TRY(gemm(c, c->pre_conv, ptr(qb), S, n, bufA, n, dt, g2, s));
        }
    }
    // ---- transformer ----
    void* x = bufA; void* nrm = bufB;
    if (k.io_proj) { if (run) TRY(gemm(c, c->in_proj, bufA, S, n, bufB, n, dt, GemmOpts{}, s)); x = bufB; nrm = bufA; }
    const int qd = k.num_attention_heads * k.head_dim, kvd = k.num_key_value_heads * k.head_dim;
    const size_t o_att = alloc(fr * qd * es), o_mlp = alloc(fr * k.intermediate_size * es);
    for (int l = 0; l < k.num_hidden_layers; ++l) {
        const Layer& y = c->layers[l];
        // The layer's keys and values of the window - 1 frames before this push stay where they are, in a ring buffer inside
        // the slot's state record that the attention kernel reads directly; the new rows are appended to it after the push
        // (the second stream_carry).  The work tensor holds the new rows only.
        const int ring_rows = k.sliding_window - 1;
        SBuf qkv{alloc((size_t)S * n * (qd + 2 * kvd) * es), 0, n, qd + 2 * kvd, es};
        const long long ring_off = st_off;
        if (ring_rows > 0) {
            Q3CarryDesc d{};
            d.work_off = (long long)qkv.off; d.work_bstride = (long long)n * qkv.ch * es; d.state_off = st_off;
            d.work_ld = (int)(qkv.ch * es); d.col_off = (int)(qd * es); d.row_bytes = (int)(2 * kvd * es);
            d.rows = ring_rows; d.new_rows = n; d.ring = 1;
            if (descs) descs->push_back(d);
            st_off += (long long)align_up((size_t)d.rows * d.row_bytes, 16);
        }
        if (!run) continue;
        LAUNCH_B(c, s, "rmsnorm", 2 * fr * H * es, q3_launch_rmsnorm(x, y.ln1, nrm, (long long)fr, H, k.rms_norm_eps, dt, s));
        TRY(gemm(c, y.qkv, nrm, S, n, ptr(qkv), n, dt, GemmOpts{}, s));
        Q3KvRing ring{P->state, P->slot_stride, ring_off, slots, P->capacity};
        LAUNCH_B(c, s, "swa_attention", fr * (size_t)(2 * qd + 2 * kvd) * es,
                 q3_launch_swa_attention(ptr(qkv), W(o_att), c->cos_t, c->sin_t, S, n, k.num_attention_heads, k.num_key_value_heads,
                                         k.head_dim, k.sliding_window, dt, s, ring_rows, pos0, k.max_positions, &ring, c->tuning));
        GemmOpts o1; o1.scale = y.ls_attn; o1.R = x;
        TRY(gemm(c, y.o, W(o_att), S, n, x, n, dt, o1, s));
        LAUNCH_B(c, s, "rmsnorm", 2 * fr * H * es, q3_launch_rmsnorm(x, y.ln2, nrm, (long long)fr, H, k.rms_norm_eps, dt, s));
        GemmOpts o2; o2.act = Q3_ACT_SWIGLU;
        TRY(gemm(c, y.gu, nrm, S, n, W(o_mlp), n, dt, o2, s));
        GemmOpts o3; o3.scale = y.ls_mlp; o3.R = x;
        TRY(gemm(c, y.down, W(o_mlp), S, n, x, n, dt, o3, s));
    }
    if (run) LAUNCH_B(c, s, "rmsnorm", 2 * fr * H * es, q3_launch_rmsnorm(x, c->final_norm, nrm, (long long)fr, H, k.rms_norm_eps, dt, s));
    void* cur;
    if (k.io_proj) { if (run) TRY(gemm(c, c->out_proj, nrm, S, n, x, n, dt, GemmOpts{}, s)); cur = x; }
    else cur = nrm;

    // ---- upsample stages ----
    int L = n;
    SBuf conv_in_x{};
    const int nu = k.num_upsampling_ratios;
    if (nu == 0) {
        conv_in_x = tensor(6, L, Lt, es);
        if (run) HIPCHK(c, hipMemcpy2DAsync((char*)ptr(conv_in_x) + new_rows_off(conv_in_x) * es, item_stride(conv_in_x) * es, cur,
                                            (size_t)L * Lt * es, (size_t)L * Lt * es, S, hipMemcpyDeviceToDevice, s));
    }
    for (int i = 0; i < nu; ++i) {
        const UpStage& u = c->ups[i];
        const int Lf = L * u.f;
        SBuf d_in = tensor(6, Lf, Lt, es);                       // the stage's transposed-conv output: depthwise conv looks back 6 rows
        const size_t o_d = alloc((size_t)S * Lf * Lt * es), o_pw = alloc((size_t)S * Lf * 4 * Lt * es);
        const bool last = i + 1 == nu;
        size_t o_next = 0;
        if (last) conv_in_x = tensor(6, Lf, Lt, es); else o_next = alloc((size_t)S * Lf * Lt * es);
        if (run) {
            GemmOpts g0; g0.segmented = true; g0.y_off = new_rows_off(d_in); g0.y_bstride = item_stride(d_in);
            TRY(gemm(c, u.tconv, cur, S, L, ptr(d_in), L, dt, g0, s));
            LAUNCH_B(c, s, "dwconv_ln", 2 * (size_t)S * Lf * Lt * es,
                     q3_launch_dwconv_ln(ptr(d_in), u.dw_w, u.dw_b, u.ln_w, u.ln_b, W(o_d), S, Lf, Lt, 7, k.convnext_eps, dt, s, d_in.hist));
            GemmOpts g1; g1.act = Q3_ACT_GELU;
            TRY(gemm(c, u.pw1, W(o_d), S, Lf, W(o_pw), Lf, dt, g1, s));
            GemmOpts g2; g2.segmented = true; g2.scale = u.gamma;
            g2.R = ptr(d_in); g2.r_off = new_rows_off(d_in); g2.r_bstride = item_stride(d_in);
            if (last) { g2.y_off = new_rows_off(conv_in_x); g2.y_bstride = item_stride(conv_in_x); }
            TRY(gemm(c, u.pw2, W(o_pw), S, Lf, last ? ptr(conv_in_x) : (void*)W(o_next), Lf, dt, g2, s));
        }
        cur = W(o_next);
        L = Lf;
    }

    // ---- conv decoder ----
    SBuf t_in = tensor(1, L, k.decoder_dim, es);                  // input of block 0's transposed conv (its SnakeBeta applied)
    if (run) {
        GemmOpts g; g.segmented = true; g.x_hist = conv_in_x.hist; g.post = &c->blocks[0].head;
        g.y_off = new_rows_off(t_in); g.y_bstride = item_stride(t_in);
        TRY(gemm(c, c->conv_in, ptr(conv_in_x), S, L, ptr(t_in), L, dt, g, s));
    }
    int ch = k.decoder_dim;
    for (int i = 0; i < nb; ++i) {
        const Block& b = c->blocks[i];
        ch /= 2;
        const int Lr = L * b.rate;
        const int* rs_t = rs_all + (size_t)(2 * i) * S;
        const int* rs_u = rs_all + (size_t)(2 * i + 1) * S;
        const bool fuse = c->fuse_units && q3_convgemm_fusable(ch, dt, c->tuning);
        // unfused units keep a raw residual stream beside the snaked tensors and an h tensor between their two convs;
        // fused units read and write the raw stream only (their [carried | new] tensors ARE the residual stream)
        const size_t o_raw = fuse ? 0 : alloc((size_t)S * Lr * ch * es);
        const size_t o_h = fuse ? 0 : alloc((size_t)S * Lr * ch * es);
        void* raw = W(o_raw);
        SBuf u_in = tensor(6 * k.dilations[0], Lr, ch, es);
        if (run) {
            // [previous | new] input rows: group j of r output rows needs rows j-1 and j
            GemmOpts g; g.segmented = true; g.x_hist = 1; g.off_shift = -1; g.row_start = rs_t;
            if (fuse) {
                g.y_off = new_rows_off(u_in); g.y_bstride = item_stride(u_in);
                TRY(gemm(c, b.tconv, ptr(t_in), S, L, ptr(u_in), L, dt, g, s));
            } else {
                g.post = &b.first_act_tiled;
                g.Y2 = ptr(u_in); g.y2_off = new_rows_off(u_in); g.y2_bstride = item_stride(u_in);
                TRY(gemm(c, b.tconv, ptr(t_in), S, L, raw, L, dt, g, s));
            }
        }
        L = Lr;
        for (size_t j = 0; j < b.units.size(); ++j) {
            const Unit& u = b.units[j];
            const bool last = j + 1 == b.units.size();
            const Snake* post = !last ? &b.units[j + 1].a1 : ((i + 1 < nb) ? &c->blocks[i + 1].head : &c->final_snake);
            SBuf next = last ? tensor(i + 1 < nb ? 1 : 6, L, ch, es) : tensor(6 * k.dilations[j + 1], L, ch, es);
            if (run) {
                if (fuse) {
                    GemmOpts g; g.segmented = true; g.row_start = rs_u; g.x_hist = u_in.hist;
                    g.act = Q3_ACT_SNAKE; g.epi = &u.a2; g.second = &u.c2; g.pre = &u.a1;
                    g.R = ptr(u_in); g.r_off = new_rows_off(u_in); g.r_bstride = item_stride(u_in);
                    g.y_off = new_rows_off(next); g.y_bstride = item_stride(next);
                    if (last) g.post = post;
                    TRY(gemm(c, u.c1, ptr(u_in), S, L, ptr(next), L, dt, g, s));
                } else {
                    GemmOpts g; g.segmented = true; g.R = raw; g.post = post; g.row_start = rs_u;
                    void* Y;
                    if (last) { Y = ptr(next); g.y_off = new_rows_off(next); g.y_bstride = item_stride(next); }
                    else { Y = raw; g.Y2 = ptr(next); g.y2_off = new_rows_off(next); g.y2_bstride = item_stride(next); }
                    GemmOpts g1; g1.segmented = true; g1.x_hist = u_in.hist; g1.act = Q3_ACT_SNAKE; g1.epi = &u.a2;
                    TRY(gemm(c, u.c1, ptr(u_in), S, L, W(o_h), L, dt, g1, s));
                    TRY(gemm(c, u.c2, W(o_h), S, L, Y, L, dt, g, s));
                }
            }
            u_in = next;
        }
        t_in = u_in;
    }
    if (run) {   // final conv to one channel, clamp fused, fp32 samples out
        GemmOpts g; g.segmented = true; g.x_hist = t_in.hist; g.act = Q3_ACT_CLAMP; g.row_start = rs_all + (size_t)(2 * nb - 1) * S;
        TRY(gemm(c, c->conv_out, ptr(t_in), S, L, wav, L, Q3_F32, g, s));
        LAUNCH_B(c, s, "stream_carry", 2.0 * S * (plan->carried_bytes + plan->appended_bytes),
                 q3_launch_stream_carry(plan->descs, plan->ndesc, ws, P->state, P->slot_stride, slots, S, P->capacity, 1, pos0, s));
    }
    if (ws_total) *ws_total = off;
    if (state_total) *state_total = st_off;
    return 0;
}

int stream_plan(q3c_stream_pool* P, int S, int n, StreamPlan** out, bool create) {
    q3c_ctx* c = P->c;
    auto it = P->plans.find({S, n});
    if (it == P->plans.end() && !create)
        return fail(c, Q3C_ERR_STATE, "no plan for this (streams, frames) pair: call q3c_stream_workspace_bytes(streams, frames) first "
                                      "(it sizes the workspace and allocates the push's descriptor table; q3c_stream_push allocates nothing)");
    if (it == P->plans.end()) {
        std::vector<Q3CarryDesc> descs;
        StreamPlan pl;
        long long st = 0;
        TRY(stream_sequence(P, S, n, false, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &descs, &pl.ws_bytes, &st));
        if (st > P->slot_stride) return fail(c, Q3C_ERR_STATE, "stream state record size changed");
        void* p = nullptr;
        HIPCHK(c, hipMalloc(&p, descs.size() * sizeof(Q3CarryDesc)));
        P->allocs.push_back(p);
        HIPCHK(c, hipMemcpy(p, descs.data(), descs.size() * sizeof(Q3CarryDesc), hipMemcpyHostToDevice));
        pl.descs = (Q3CarryDesc*)p; pl.ndesc = (int)descs.size();
        for (const Q3CarryDesc& d : descs) {
            if (d.ring) pl.appended_bytes += (double)std::min(d.rows, d.new_rows) * d.row_bytes;
            else pl.carried_bytes += (double)d.rows * d.row_bytes;
        }
        it = P->plans.emplace(std::make_pair(S, n), pl).first;
    }
    *out = &it->second;
    return 0;
}

}  // namespace

// =========================================================================================================
extern "C" {

const char* q3c_version(void) { return "q3c 0.2 (gfx950)"; }

int q3c_create(const q3c_config* cfg, q3c_ctx** out) {
    if (!cfg || !out) return Q3C_ERR_INVALID_ARGUMENT;
    q3c_ctx* c = new q3c_ctx();
    if (cfg->struct_size == (int)sizeof(q3c_config)) c->cfg = *cfg; else c->cfg.struct_size = cfg->struct_size;
    *out = c;                       // returned even on failure so the caller can read q3c_last_error
    int rc = validate_config(c);
    if (rc) return rc;
    c->dtype = cfg->precision == Q3C_PRECISION_BF16 ? Q3_BF16 : Q3_F32;
    c->total_upsample = 1;
    for (int i = 0; i < cfg->num_upsampling_ratios; ++i) c->total_upsample *= cfg->upsampling_ratios[i];
    int later = 1;
    c->trim = 0;
    for (int i = cfg->num_upsample_rates - 1; i >= 0; --i) { later *= cfg->upsample_rates[i]; c->trim += later; }
    c->total_upsample *= later;
    c->G = cfg->frontend == Q3C_FRONTEND_SPLIT_RVQ ? (cfg->num_quantizers > cfg->num_semantic_quantizers ? 2 : 1) : 0;
    return 0;
}

void q3c_destroy(q3c_ctx* c) {
    if (!c) return;
    for (auto& r : c->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
}

const char* q3c_last_error(const q3c_ctx* c) { return c ? c->err.c_str() : "null context"; }

int q3c_load_weight(q3c_ctx* c, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    if (!c) return Q3C_ERR_INVALID_ARGUMENT;
    if (!name || !data || !shape || ndim < 1 || ndim > 4) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_load_weight: bad argument");
    if (c->finalized) return fail(c, Q3C_ERR_STATE, "weights are frozen after q3c_finalize");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { if (shape[i] < 0) return fail(c, Q3C_ERR_WEIGHT, "negative dimension"); t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(data, data + n);
    c->staged[name] = std::move(t);
    return 0;
}

int q3c_finalize(q3c_ctx* c) {
    if (!c) return Q3C_ERR_INVALID_ARGUMENT;
    if (c->finalized) return fail(c, Q3C_ERR_STATE, "already finalized");
    const q3c_config& k = c->cfg;
    const int dt = c->dtype;
    int rc = 0;
    void* p;
    TRY(dev_alloc(c, 16, &p));
    c->err_flag = (int*)p;
    HIPCHK(c, hipMemset(c->err_flag, 0, 16));
    TRY(ensure_zero_page(c));

    const int H = k.hidden_size, Lt = k.latent_dim;
    // ---- front end ----
    if (k.frontend == Q3C_FRONTEND_MEAN_EMBEDDING) {
        GETW(e, "code_embedding.weight", (int64_t)k.codebook_size * k.num_quantizers, H);
        TRY(upload_f32(c, e->data, &c->emb_table));
    } else {
        const int ns = k.num_semantic_quantizers, D = k.codebook_dim;
        std::vector<int> gs = {0, ns};
        if (c->G == 2) gs.push_back(k.num_quantizers);
        std::vector<float> proj((size_t)k.rvq_out_dim * c->G * D);
        std::vector<void*> tables_t;
        for (int g = 0; g < c->G; ++g) {
            const std::string gn = g == 0 ? "semantic" : "acoustic";
            for (int i = 0; i < gs[g + 1] - gs[g]; ++i) {
                const std::string base = "quantizer." + gn + ".layers." + std::to_string(i) + ".";
                GETW(es, base + "embed_sum", k.codebook_size, D);
                GETW(cu, base + "cluster_usage", k.codebook_size);
                std::vector<float> tab((size_t)k.codebook_size * D);
                for (int r = 0; r < k.codebook_size; ++r) {
                    const float den = cu->data[r] < k.rvq_eps ? k.rvq_eps : cu->data[r];      // clamp(min = eps)
                    for (int d = 0; d < D; ++d) tab[(size_t)r * D + d] = es->data[(size_t)r * D + d] / den;   // IEEE fp32 divide
                }
                float* tp; TRY(upload_f32(c, tab, &tp));
                c->tables.push_back(tp);
                {   // the encode direction reads the table by dimension: [D/4][bins][4], then the rows' squared norms [bins]
                    std::vector<float> tt(tab.size() + k.codebook_size);
                    for (int r = 0; r < k.codebook_size; ++r) {
                        double n2 = 0.0;
                        for (int d = 0; d < D; ++d) {
                            tt[((size_t)(d >> 2) * k.codebook_size + r) * 4 + (d & 3)] = tab[(size_t)r * D + d];
                            n2 += (double)tab[(size_t)r * D + d] * tab[(size_t)r * D + d];
                        }
                        tt[tab.size() + r] = (float)n2;
                    }
                    float* tpt; TRY(upload_f32(c, tt, &tpt));
                    tables_t.push_back(tpt);
                }
            }
            GETW(op, "quantizer." + gn + ".output_proj.weight", k.rvq_out_dim, D, 1);
            for (int n = 0; n < k.rvq_out_dim; ++n)
                for (int d = 0; d < D; ++d) proj[((size_t)n * c->G + g) * D + d] = op->data[(size_t)n * D + d];
        }
        TRY(dev_alloc(c, c->tables.size() * sizeof(float*), &p));
        c->tables_dev = (float**)p;
        HIPCHK(c, hipMemcpy(p, c->tables.data(), c->tables.size() * sizeof(float*), hipMemcpyHostToDevice));
        TRY(dev_alloc(c, tables_t.size() * sizeof(void*), &p));
        c->tables_t_dev = (void**)p;
        HIPCHK(c, hipMemcpy(p, tables_t.data(), tables_t.size() * sizeof(void*), hipMemcpyHostToDevice));
        TRY(dev_alloc(c, gs.size() * sizeof(int), &p));
        c->group_start_dev = (int*)p;
        HIPCHK(c, hipMemcpy(p, gs.data(), gs.size() * sizeof(int), hipMemcpyHostToDevice));
        // encode direction (optional): the groups' bias-free 1x1 input_proj [codebook_dim][in_dim][1], all or none
        if (c->staged.count("quantizer.semantic.input_proj.weight")) {
            const HostTensor& first = c->staged["quantizer.semantic.input_proj.weight"];
            if (first.shape.size() != 3 || first.shape[0] != D || first.shape[2] != 1 || first.shape[1] < 1)
                return fail(c, Q3C_ERR_WEIGHT, "quantizer.semantic.input_proj.weight must be [codebook_dim][in_dim][1]");
            const int in_dim = (int)first.shape[1];
            std::vector<float> ip((size_t)c->G * D * in_dim);
            for (int g = 0; g < c->G; ++g) {
                GETW(w_in, std::string("quantizer.") + (g == 0 ? "semantic" : "acoustic") + ".input_proj.weight", D, in_dim, 1);
                std::memcpy(&ip[(size_t)g * D * in_dim], w_in->data.data(), (size_t)D * in_dim * sizeof(float));
            }
            TRY(upload_f32(c, ip, &c->rvq_in_proj));
            c->rvq_in_dim = in_dim;
        }
        // both groups' 1x1 projections as ONE contraction over the concatenated sums: always fp32 operands
        c->rvq_proj.taps = 1; c->rvq_proj.N = k.rvq_out_dim; c->rvq_proj.K = c->G * D; c->rvq_proj.in_dtype = Q3_F32;
        TRY(upload_typed(c, proj, Q3_F32, &c->rvq_proj.w));
        TRY(pack_conv(c, "pre_conv.conv", Lt, k.rvq_out_dim, k.pre_conv_kernel, 1, dt, &c->pre_conv));
    }
    if (k.io_proj) {
        TRY(pack_linear(c, "pre_transformer.input_proj.weight", "pre_transformer.input_proj.bias", H, Lt, dt, &c->in_proj));
        TRY(pack_linear(c, "pre_transformer.output_proj.weight", "pre_transformer.output_proj.bias", Lt, H, dt, &c->out_proj));
    }
    // ---- transformer ----
    const int qd = k.num_attention_heads * k.head_dim, kvd = k.num_key_value_heads * k.head_dim, I = k.intermediate_size;
    c->layers.resize(k.num_hidden_layers);
    for (int l = 0; l < k.num_hidden_layers; ++l) {
        Layer& y = c->layers[l];
        const std::string pre = "pre_transformer.layers." + std::to_string(l) + ".";
        TRY(pack_vec(c, pre + "input_layernorm.weight", H, &y.ln1));
        TRY(pack_vec(c, pre + "post_attention_layernorm.weight", H, &y.ln2));
        TRY(pack_vec(c, pre + "self_attn_layer_scale.scale", H, &y.ls_attn));
        TRY(pack_vec(c, pre + "mlp_layer_scale.scale", H, &y.ls_mlp));
        GETW(wq, pre + "self_attn.q_proj.weight", qd, H);
        GETW(wk, pre + "self_attn.k_proj.weight", kvd, H);
        GETW(wv, pre + "self_attn.v_proj.weight", kvd, H);
        std::vector<float> cat;
        cat.reserve((size_t)(qd + 2 * kvd) * H);
        cat.insert(cat.end(), wq->data.begin(), wq->data.end());
        cat.insert(cat.end(), wk->data.begin(), wk->data.end());
        cat.insert(cat.end(), wv->data.begin(), wv->data.end());
        y.qkv.taps = 1; y.qkv.N = qd + 2 * kvd; y.qkv.K = H; y.qkv.in_dtype = dt;
        TRY(upload_typed(c, cat, dt, &y.qkv.w));
        TRY(pack_linear(c, pre + "self_attn.o_proj.weight", "", H, qd, dt, &y.o));
        GETW(wg, pre + "mlp.gate_proj.weight", I, H);
        GETW(wu, pre + "mlp.up_proj.weight", I, H);
        std::vector<float> gu((size_t)2 * I * H);     // 32-row blocks alternate gate / up so one lane owns both halves
        for (int blk = 0; blk < I / 32; ++blk)
            for (int r = 0; r < 32; ++r) {
                std::memcpy(&gu[((size_t)(2 * blk) * 32 + r) * H], &wg->data[((size_t)blk * 32 + r) * H], (size_t)H * 4);
                std::memcpy(&gu[((size_t)(2 * blk + 1) * 32 + r) * H], &wu->data[((size_t)blk * 32 + r) * H], (size_t)H * 4);
            }
        y.gu.taps = 1; y.gu.N = 2 * I; y.gu.K = H; y.gu.in_dtype = dt;
        TRY(upload_typed(c, gu, dt, &y.gu.w));
        TRY(pack_linear(c, pre + "mlp.down_proj.weight", "", H, I, dt, &y.down));
    }
    TRY(pack_vec(c, "pre_transformer.norm.weight", H, &c->final_norm));
    {   // rotary tables: freq = float(pos) * inv_freq in fp32, then cos / sin of that fp32 angle
        const int half = k.head_dim / 2;
        GETW(inv, "rope.inv_freq", half);
        std::vector<float> ct((size_t)k.max_positions * half), st(ct.size());
        for (int pos = 0; pos < k.max_positions; ++pos)
            for (int e = 0; e < half; ++e) {
                const float ang = (float)pos * inv->data[e];
                ct[(size_t)pos * half + e] = (float)std::cos((double)ang);
                st[(size_t)pos * half + e] = (float)std::sin((double)ang);
            }
        TRY(upload_f32(c, ct, &c->cos_t));
        TRY(upload_f32(c, st, &c->sin_t));
    }
    // ---- upsample stages ----
    c->ups.resize(k.num_upsampling_ratios);
    for (int i = 0; i < k.num_upsampling_ratios; ++i) {
        UpStage& u = c->ups[i];
        u.f = k.upsampling_ratios[i];
        const std::string pre = "upsample." + std::to_string(i) + ".";
        TRY(pack_tconv(c, pre + "0.conv", Lt, Lt, u.f, u.f, dt, &u.tconv));
        GETW(dw, pre + "1.dwconv.conv.weight", Lt, 1, 7);
        std::vector<float> dwt((size_t)7 * Lt);
        for (int ch = 0; ch < Lt; ++ch)
            for (int j = 0; j < 7; ++j) dwt[(size_t)j * Lt + ch] = dw->data[(size_t)ch * 7 + j];
        TRY(upload_f32(c, dwt, &u.dw_w));
        TRY(pack_vec(c, pre + "1.dwconv.conv.bias", Lt, &u.dw_b));
        TRY(pack_vec(c, pre + "1.norm.weight", Lt, &u.ln_w));
        TRY(pack_vec(c, pre + "1.norm.bias", Lt, &u.ln_b));
        TRY(pack_linear(c, pre + "1.pwconv1.weight", pre + "1.pwconv1.bias", 4 * Lt, Lt, dt, &u.pw1));
        TRY(pack_linear(c, pre + "1.pwconv2.weight", pre + "1.pwconv2.bias", Lt, 4 * Lt, dt, &u.pw2));
        TRY(pack_vec(c, pre + "1.gamma", Lt, &u.gamma));
    }
    // ---- conv decoder ----
    TRY(pack_conv(c, "decoder.0.conv", k.decoder_dim, Lt, 7, 1, dt, &c->conv_in));
    c->blocks.resize(k.num_upsample_rates);
    int ch = k.decoder_dim;
    for (int i = 0; i < k.num_upsample_rates; ++i) {
        Block& b = c->blocks[i];
        const int cin = ch, cout = ch / 2, r = k.upsample_rates[i];
        b.rate = r;
        const std::string pre = "decoder." + std::to_string(i + 1) + ".block.";
        TRY(pack_snake(c, pre + "0.", cin, &b.head));
        TRY(pack_tconv(c, pre + "1.conv", cin, cout, 2 * r, r, dt, &b.tconv));
        b.units.resize(k.num_dilations);
        for (int j = 0; j < k.num_dilations; ++j) {
            Unit& u = b.units[j];
            const std::string q = pre + std::to_string(j + 2) + ".";
            TRY(pack_snake(c, q + "act1.", cout, &u.a1));
            TRY(pack_conv(c, q + "conv1.conv", cout, cout, 7, k.dilations[j], dt, &u.c1));
            TRY(pack_snake(c, q + "act2.", cout, &u.a2));
            TRY(pack_conv(c, q + "conv2.conv", cout, cout, 1, 1, dt, &u.c2));
        }
        TRY(pack_snake(c, pre + "2.act1.", cout, &b.first_act_tiled, r));
        ch = cout;
    }
    const int nb = k.num_upsample_rates;
    TRY(pack_snake(c, "decoder." + std::to_string(nb + 1) + ".", ch, &c->final_snake));
    TRY(pack_conv(c, "decoder." + std::to_string(nb + 2) + ".conv", 1, ch, 7, 1, dt, &c->conv_out));
    if (dt == Q3_BF16) {
        int rc = 0;
        GETW(wf, "decoder." + std::to_string(nb + 2) + ".conv.weight", 1, ch, 7);
        std::vector<float> p16((size_t)16 * ch, 0.f);
        for (int ci = 0; ci < ch; ++ci)
            for (int t = 0; t < 7; ++t) p16[(size_t)t * ch + ci] = wf->data[(size_t)ci * 7 + t];
        TRY(upload_typed(c, p16, Q3_BF16, &c->conv_out_fold));
    }
    if (c->enc.configured) TRY(encoder_finalize(c));
    HIPCHK(c, hipGetDevice(&c->device));
    {   // this device's copy of the static workspace (every GPU has its own, at its own address)
        void* sp = nullptr;
        HIPCHK(c, hipGetSymbolAddress(&sp, HIP_SYMBOL(q3c_static_scratch)));
        if (!sp || ((uintptr_t)sp & 255)) return fail(c, Q3C_ERR_HIP, "static workspace has no usable device address");
        c->static_scratch = sp;
    }
    c->staged.clear();
    c->finalized = true;
    return 0;
}

int q3c_output_length(const q3c_ctx* c, int32_t frames, int64_t* samples) {
    if (!c || !samples || frames < 1) return Q3C_ERR_INVALID_ARGUMENT;
    *samples = (int64_t)frames * c->total_upsample - c->trim;
    return 0;
}

int q3c_workspace_bytes(q3c_ctx* c, int32_t B, int32_t T, size_t* bytes) {
    if (!c || !bytes) return Q3C_ERR_INVALID_ARGUMENT;
    if (B < 1 || T < 1) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "batch and frames must be >= 1");
    *bytes = make_layout(c, B, T).total;
    return 0;
}

int q3c_set_tap(q3c_ctx* c, int32_t stage, float* buf, size_t cap) {
    if (!c) return Q3C_ERR_INVALID_ARGUMENT;
    if (!buf) c->taps.erase(stage); else c->taps[stage] = {buf, cap};
    return 0;
}

int q3c_check_errors(q3c_ctx* c, void* stream) {
    if (!c || !c->finalized) return Q3C_ERR_STATE;
    int flag = 0;
    HIPCHK(c, hipMemcpyAsync(&flag, c->err_flag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(c, hipStreamSynchronize((hipStream_t)stream));
    if (flag) {
        HIPCHK(c, hipMemsetAsync(c->err_flag, 0, sizeof(int), (hipStream_t)stream));
        if (flag & 1) return fail(c, Q3C_ERR_CODES, "a code index was outside [0, codebook_size)");
        if (flag & 2) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "a stream slot id was outside the pool");
        return fail(c, Q3C_ERR_INVALID_ARGUMENT, "a stream ran past max_positions frames");
    }
    return 0;
}

int q3c_rvq_dequant_f32(q3c_ctx* c, const int64_t* codes, int32_t B, int32_t T, float* sums, void* stream) {
    if (!c) return Q3C_ERR_INVALID_ARGUMENT;
    if (!c->finalized) return fail(c, Q3C_ERR_STATE, "q3c_finalize has not run");
    if (c->cfg.frontend != Q3C_FRONTEND_SPLIT_RVQ) return fail(c, Q3C_ERR_STATE, "context has no RVQ front end");
    if (!codes || !sums || B < 1 || T < 1) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_rvq_dequant_f32: bad argument");
    LAUNCH(c, q3_launch_rvq_gather_sum((const long long*)codes, B, c->cfg.num_quantizers, T, c->cfg.codebook_size, c->cfg.codebook_dim,
                                       c->tables_dev, c->group_start_dev, c->G, sums, c->err_flag, (hipStream_t)stream));
    return 0;
}

static int forward_impl(q3c_ctx* c, const int64_t* codes, int32_t B, int32_t T, float* wav, void* ws, size_t ws_bytes, void* stream,
                        int32_t skip_frames = 0, const int64_t* spans = nullptr);

int q3c_rvq_encode(q3c_ctx* c, const float* emb, int32_t B, int32_t T, int32_t in_dim, int64_t* codes, void* stream) {
    if (!c) return Q3C_ERR_INVALID_ARGUMENT;
    if (!c->finalized) return fail(c, Q3C_ERR_STATE, "q3c_finalize has not run");
    if (c->cfg.frontend != Q3C_FRONTEND_SPLIT_RVQ) return fail(c, Q3C_ERR_STATE, "context has no RVQ front end");
    if (!emb || !codes || B < 1 || T < 1) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_rvq_encode: bad argument");
    const int want = c->rvq_in_proj ? c->rvq_in_dim : c->cfg.codebook_dim;
    if (in_dim != want)
        return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_rvq_encode: embeddings have " + std::to_string(in_dim) + " channels, the quantizer takes " +
                                                     std::to_string(want) + (c->rvq_in_proj ? " (its input_proj)" : " (no input_proj was loaded)"));
    LAUNCH(c, q3_launch_rvq_encode(emb, B, T, in_dim, c->cfg.num_quantizers, c->cfg.codebook_size, c->cfg.codebook_dim, c->tables_dev,
                                   (const void* const*)c->tables_t_dev, c->group_start_dev, c->G, c->rvq_in_proj, (long long*)codes,
                                   (hipStream_t)stream, c->tuning));
    return 0;
}

int q3c_forward(q3c_ctx* c, const int64_t* codes, int32_t B, int32_t T, float* wav, void* ws, size_t ws_bytes, void* stream) {
    if (!c) return Q3C_ERR_INVALID_ARGUMENT;
    if (!c->finalized) return fail(c, Q3C_ERR_STATE, "q3c_finalize has not run");
    if (!codes || !wav || !ws || B < 1 || T < 1) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_forward: bad argument");
    return forward_impl(c, codes, B, T, wav, ws, ws_bytes, stream);
}

int q3c_forward_skip(q3c_ctx* c, const int64_t* codes, int32_t B, int32_t T, int32_t skip_frames, float* wav, void* ws, size_t ws_bytes,
                     void* stream) {
    if (!c) return Q3C_ERR_INVALID_ARGUMENT;
    if (!c->finalized) return fail(c, Q3C_ERR_STATE, "q3c_finalize has not run");
    if (!codes || !wav || !ws || B < 1 || T < 1) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_forward_skip: bad argument");
    if (skip_frames < 0 || skip_frames >= T) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_forward_skip: skip_frames must be in [0, frames)");
    return forward_impl(c, codes, B, T, wav, ws, ws_bytes, stream, skip_frames);
}

// q3c_forward on the library's static workspace: no device memory is allocated or freed here or in anything it calls
// (launches, events and taps included: profiling, the one consumer of hipEventCreate, is refused).
int q3c_forward_static(q3c_ctx* c, const int64_t* codes, int32_t B, int32_t T, float* wav, void* stream) {
    if (!c) return Q3C_ERR_INVALID_ARGUMENT;
    if (!c->finalized) return fail(c, Q3C_ERR_STATE, "q3c_finalize has not run");
    if (!codes || !wav || B < 1 || T < 1) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_forward_static: bad argument");
    if (c->profiling) return fail(c, Q3C_ERR_STATE, "q3c_forward_static: per-kernel profiling creates events; disable it first");
    int dev = -1;
    HIPCHK(c, hipGetDevice(&dev));
    if (dev != c->device) return fail(c, Q3C_ERR_STATE, "q3c_forward_static: the context's device is not current");
    const size_t need = make_layout(c, B, T).total;
    if (need > (size_t)Q3C_STATIC_SCRATCH_BYTES)
        return fail(c, Q3C_ERR_WORKSPACE, "q3c_forward_static: batch x frames needs " + std::to_string(need) + " bytes of workspace, the static one holds " +
                                              std::to_string((size_t)Q3C_STATIC_SCRATCH_BYTES) + "; use q3c_forward with a caller workspace");
    return forward_impl(c, codes, B, T, wav, c->static_scratch, (size_t)Q3C_STATIC_SCRATCH_BYTES, stream);
}

int q3c_forward_scatter(q3c_ctx* c, const int64_t* codes, int32_t B, int32_t T, int32_t skip_frames, const int64_t* spans, float* out,
                        void* ws, size_t ws_bytes, void* stream) {
    if (!c) return Q3C_ERR_INVALID_ARGUMENT;
    if (!c->finalized) return fail(c, Q3C_ERR_STATE, "q3c_finalize has not run");
    if (!codes || !out || !spans || !ws || B < 1 || T < 1) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_forward_scatter: bad argument");
    if (skip_frames < 0 || skip_frames >= T) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_forward_scatter: skip_frames must be in [0, frames)");
    return forward_impl(c, codes, B, T, out, ws, ws_bytes, stream, skip_frames, spans);
}

static int forward_impl(q3c_ctx* c, const int64_t* codes, int32_t B, int32_t T, float* wav, void* ws, size_t ws_bytes, void* stream,
                        int32_t skip_frames, const int64_t* spans) {
    const q3c_config& k = c->cfg;
    if (T > k.max_positions) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "frames exceed max_positions");
    if ((uintptr_t)ws & 255) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
    const Layout lay = make_layout(c, B, T);
    if (ws_bytes < lay.total) return fail(c, Q3C_ERR_WORKSPACE, "workspace too small: need " + std::to_string(lay.total));
    hipStream_t s = (hipStream_t)stream;
    const int dt = c->dtype;
    const size_t es = esize(dt);
    char* base = (char*)ws;
    float* S = (float*)(base + lay.oS);
    void* Qb = base + lay.oQ;
    void* bufA = base + lay.oA;
    void* bufB = base + lay.oB;
    char* bufC = base + lay.oC;
    const size_t fr = (size_t)B * T;
    const int H = k.hidden_size, Lt = k.latent_dim;

    // ---- front end -> x0 [B,T,Lt] in bufA ----
    if (k.frontend == Q3C_FRONTEND_MEAN_EMBEDDING) {
        LAUNCH_B(c, s, "mean_embedding", fr * ((size_t)k.num_quantizers * (8 + H * 4) + H * es),
                 q3_launch_mean_embedding((const long long*)codes, B, k.num_quantizers, T, k.codebook_size, H, c->emb_table, bufA, dt,
                                          c->err_flag, s));
    } else {
        LAUNCH_B(c, s, "rvq_gather_sum", fr * ((size_t)k.num_quantizers * (8 + k.codebook_dim * 4) + (size_t)c->G * k.codebook_dim * 4),
                 q3_launch_rvq_gather_sum((const long long*)codes, B, k.num_quantizers, T, k.codebook_size, k.codebook_dim,
                                          c->tables_dev, c->group_start_dev, c->G, S, c->err_flag, s));
        TRY(gemm(c, c->rvq_proj, S, B, T, Qb, T, dt, GemmOpts{}, s));
        TRY(tap_out(c, Q3C_STAGE_RVQ_OUT, Qb, fr * k.rvq_out_dim, s));
        TRY(gemm(c, c->pre_conv, Qb, B, T, bufA, T, dt, GemmOpts{}, s));
    }
    TRY(tap_out(c, Q3C_STAGE_TRANSFORMER_IN, bufA, fr * Lt, s));

    // ---- transformer ----
    void* cur; void* other;
    TRY(transformer_stack(c, B, T, bufA, bufB, bufC, s, &cur, &other));
    TRY(tap_out(c, Q3C_STAGE_TRANSFORMER_OUT, cur, fr * Lt, s));

    // ---- upsample stages: transposed conv (k = s = f) then ConvNeXt ----
    int L = T;
    for (int i = 0; i < k.num_upsampling_ratios; ++i) {
        const UpStage& u = c->ups[i];
        TRY(gemm(c, u.tconv, cur, B, L, other, L, dt, GemmOpts{}, s));       // [B, L, f*Lt] == [B, L*f, Lt]
        L *= u.f;
        const size_t rows = (size_t)B * L;
        void* d = bufC;
        void* pw = bufC + align_up(rows * Lt * es);
        LAUNCH_B(c, s, "dwconv_ln", 2 * rows * Lt * es,
                 q3_launch_dwconv_ln(other, u.dw_w, u.dw_b, u.ln_w, u.ln_b, d, B, L, Lt, 7, k.convnext_eps, dt, s));
        GemmOpts g1; g1.act = Q3_ACT_GELU;
        TRY(gemm(c, u.pw1, d, B, L, pw, L, dt, g1, s));
        GemmOpts g2; g2.scale = u.gamma; g2.R = other;
        TRY(gemm(c, u.pw2, pw, B, L, other, L, dt, g2, s));
        std::swap(cur, other);
        TRY(tap_out(c, Q3C_STAGE_UPSAMPLE0 + i, cur, rows * Lt, s));
    }

    // ---- rows nobody reads ----
    // With skip_frames > 0 the caller discards the samples of the first skip_frames frames (a chunk's recomputed left context,
    // sibling :3691-3695).  Every layer of the conv decoder is causal with a finite look-back, so the rows under those samples
    // are needed only as far as a kept sample's look-back reaches: walking back from the first kept sample gives, per launch,
    // the first row that has to be computed.  A row's arithmetic does not depend on which rows are computed beside it, so the
    // kept samples are bit for bit those of the whole forward.  The frame-rate stages (attention window: 72 frames) run whole.
    const size_t nblk = c->blocks.size();
    std::vector<int> tconv_row0(nblk, 0);
    std::vector<std::vector<int>> unit_row0(nblk);
    for (size_t i = 0; i < nblk; ++i) unit_row0[i].assign(c->blocks[i].units.size(), 0);
    int out_row0 = 0, conv_in_row0 = 0;
    const int skip_in = skip_frames;
    for (const auto& t : c->taps) if (t.second.first) skip_frames = 0;       // a captured stage is captured whole
    if (skip_frames > 0) {
        long long per_frame = 1;
        for (int i = 0; i < k.num_upsampling_ratios; ++i) per_frame *= k.upsampling_ratios[i];
        for (int i = 0; i < k.num_upsample_rates; ++i) per_frame *= k.upsample_rates[i];
        long long need = (long long)skip_frames * per_frame;                 // the first kept sample
        out_row0 = (int)need;
        need = std::max(0LL, need - (long long)(c->conv_out.taps - 1) * c->conv_out.off_step);
        for (size_t i = nblk; i-- > 0;) {
            const Block& b = c->blocks[i];
            for (size_t j = b.units.size(); j-- > 0;) {
                unit_row0[i][j] = (int)need;
                need = std::max(0LL, need - (long long)(b.units[j].c1.taps - 1) * b.units[j].c1.off_step);
            }
            need /= b.rate;                   // output rows q * rate .. of the transposed conv come from its input rows q, q + 1
            tconv_row0[i] = (int)need;
        }
        conv_in_row0 = (int)need;
    }

    // ---- conv decoder ----
    {
        GemmOpts g; g.post = &c->blocks[0].head; g.row0 = conv_in_row0;
        TRY(gemm(c, c->conv_in, cur, B, L, other, L, dt, g, s));
        std::swap(cur, other);
        TRY(tap_out(c, Q3C_STAGE_CONV_IN, cur, (size_t)B * L * k.decoder_dim, s));
    }
    // Each block: transposed conv, then residual units.  Every tensor a convolution reads is stored already passed
    // through that convolution's SnakeBeta by whichever kernel produced it (so the inputs need no transform and go
    // global -> LDS by DMA); the raw residual stream `raw` is kept beside it and updated in place.
    int ch = k.decoder_dim;
    void* free_bufs[3] = {other, bufC, base + lay.oD};
    bool folded = false;
    const int ch_last = k.decoder_dim >> c->blocks.size();
    const auto last_tap = c->taps.find(Q3C_STAGE_BLOCK0 + (int)c->blocks.size() - 1);
    // (a tap on the last block's output needs that tensor in memory; the fold needs one column tile to hold every channel)
    const bool fold_final = c->conv_out_fold && c->fuse_units && dt == Q3_BF16 && (ch_last == 96 || ch_last == 32) && c->conv_out.taps <= 8 &&
                            (last_tap == c->taps.end() || !last_tap->second.first) && !(c->tuning & 512);
    for (size_t i = 0; i < c->blocks.size(); ++i) {
        const Block& b = c->blocks[i];
        if (L < 2) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "sequence too short for the transposed convolutions");
        const int Lq = L - 1;
        ch /= 2;
        void* raw = free_bufs[0]; void* sn = free_bufs[1]; void* sp = free_bufs[2];
        const bool fuse = c->fuse_units && q3_convgemm_fusable(ch, dt, c->tuning);
        {
            // fused units read the raw stream and apply their own first SnakeBeta: the transposed conv then writes ONE tensor
            GemmOpts g; g.row0 = tconv_row0[i];
            if (!fuse) { g.post = &b.first_act_tiled; g.Y2 = sn; }
            TRY(gemm(c, b.tconv, cur, B, L, raw, Lq, dt, g, s));              // [B, L-1, r*ch] == [B, (L-1)*r, ch]
        }
        L = Lq * b.rate;
        for (size_t j = 0; j < b.units.size(); ++j) {
            const Unit& u = b.units[j];
            const bool last = j + 1 == b.units.size();
            const Snake* post = !last ? &b.units[j + 1].a1 : ((i + 1 < c->blocks.size()) ? &c->blocks[i + 1].head : &c->final_snake);
            if (fuse) {
                // snake -> conv k7 -> snake -> conv k1 -> + x in one launch on the RAW stream: one tensor in (read with a
                // halo, so the result goes to the other buffer), one tensor out; only the block's last unit applies a
                // consumer's SnakeBeta (the next block's head, or the final one), to its single output
                GemmOpts g; g.act = Q3_ACT_SNAKE; g.epi = &u.a2; g.second = &u.c2; g.pre = &u.a1; g.R = raw; g.row0 = unit_row0[i][j];
                if (last) g.post = post;
                // The decoder's last unit: its output (final SnakeBeta applied) is read by the one-channel final conv alone, so
                // it is never stored: the unit's epilogue leaves the conv's per-tap products, 8 floats per row, in `sn`.
                if (last && i + 1 == c->blocks.size() && fold_final) { g.fold_w = c->conv_out_fold; g.fold_p = (float*)sn; folded = true; }
                TRY(gemm(c, u.c1, raw, B, L, sn, L, dt, g, s));
                if (!last) std::swap(raw, sn);     // the last unit's (snaked) output stays in `sn`: the block's output
            } else {
                GemmOpts g1; g1.act = Q3_ACT_SNAKE; g1.epi = &u.a2; g1.row0 = unit_row0[i][j];
                TRY(gemm(c, u.c1, sn, B, L, sp, L, dt, g1, s));
                GemmOpts g2; g2.R = raw; g2.post = post; g2.row0 = unit_row0[i][j];
                if (last) { TRY(gemm(c, u.c2, sp, B, L, sn, L, dt, g2, s)); }
                else { g2.Y2 = sn; TRY(gemm(c, u.c2, sp, B, L, raw, L, dt, g2, s)); }
            }
        }
        // the block's output (next SnakeBeta applied) is in `sn`
        free_bufs[0] = cur; free_bufs[1] = raw; free_bufs[2] = sp;
        cur = sn;
        if (!(folded && i + 1 == c->blocks.size())) TRY(tap_out(c, Q3C_STAGE_BLOCK0 + (int)i, cur, (size_t)B * L * ch, s));
    }
    // With `spans` (q3c_forward_scatter) item b's kept samples go straight to their place in the caller's output: the folded
    // path's diagonal sum writes them there; otherwise the final conv writes a free arena and one copy launch places them.
    const int first_kept = (int)((long long)skip_in * c->total_upsample);
    if (folded) {   // the final conv's taps summed along the diagonal of the products, bias, clamp
        LAUNCH_B(c, s, "tap_sum", (size_t)B * (L - out_row0) * 36,
                 q3_launch_tap_sum((const float*)cur, c->conv_out.bias, wav, B, L, c->conv_out.taps, Q3_ACT_CLAMP, s, spans ? first_kept : out_row0,
                                   (const long long*)spans));
    } else {        // final conv to one channel, clamp fused, fp32 samples out
        GemmOpts g; g.act = Q3_ACT_CLAMP; g.row0 = out_row0;
        float* dst = spans ? (float*)free_bufs[0] : wav;
        TRY(gemm(c, c->conv_out, cur, B, L, dst, L, Q3_F32, g, s));
        if (spans) LAUNCH_B(c, s, "span_copy", (size_t)B * (L - first_kept) * 8, q3_launch_span_copy(dst, B, L, first_kept, (const long long*)spans, wav, s));
    }
    return 0;
}

// measurement aids: per context, never process-wide (two contexts in one process keep their own settings)
int q3c_debug_set_tuning(q3c_ctx* c, int32_t bits) {
    if (!c) return Q3C_ERR_INVALID_ARGUMENT;
    c->tuning = bits;
    return 0;
}

int q3c_debug_set_timeline(q3c_ctx* c, long long* dev_buffer, int32_t slots) {
    if (!c || slots < 0) return Q3C_ERR_INVALID_ARGUMENT;
    c->timeline = dev_buffer; c->timeline_slots = dev_buffer ? slots : 0;
    return 0;
}

int q3c_profile_enable(q3c_ctx* c, int32_t on) {
    if (!c) return Q3C_ERR_INVALID_ARGUMENT;
    for (auto& r : c->prof) { c->event_pool.push_back(r.a); c->event_pool.push_back(r.b); }
    c->prof.clear();
    c->profiling = on != 0;
    c->profile_detail = on == 2;      // 2: one entry per (kernel, layer shape)
    return 0;
}

int q3c_profile_read(q3c_ctx* c, q3c_profile_entry* out, int32_t cap, int32_t* count) {
    if (!c || !out || !count || cap < 1) return Q3C_ERR_INVALID_ARGUMENT;
    std::map<std::string, q3c_profile_entry> agg;
    for (auto& r : c->prof) {
        HIPCHK(c, hipEventSynchronize(r.b));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, r.a, r.b));
        q3c_profile_entry& e = agg[r.name];
        if (!e.launches) { std::memset(&e, 0, sizeof(e)); std::snprintf(e.kernel, sizeof(e.kernel), "%s", r.name); }
        e.launches += 1; e.ms += ms; e.flops += r.flops; e.bytes += r.bytes;
    }
    int n = 0;
    for (auto& kv : agg) { if (n < cap) out[n++] = kv.second; }
    *count = n;
    return 0;
}


// ---- streaming ------------------------------------------------------------------------------------------------
int q3c_stream_pool_create(q3c_ctx* c, int32_t capacity, q3c_stream_pool** out) {
    if (!c || !out) return Q3C_ERR_INVALID_ARGUMENT;
    if (!c->finalized) return fail(c, Q3C_ERR_STATE, "q3c_finalize has not run");
    if (capacity < 1) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "stream pool capacity must be >= 1");
    q3c_stream_pool* P = new q3c_stream_pool();
    P->c = c; P->capacity = capacity;
    long long st = 0;
    int rc = stream_sequence(P, 1, 1, false, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &st);
    if (rc) { delete P; return rc; }
    P->slot_stride = (long long)align_up((size_t)st, 256);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)P->slot_stride * capacity);
    if (e == hipSuccess) { P->allocs.push_back(p); P->state = (char*)p; e = hipMemset(p, 0, (size_t)P->slot_stride * capacity); }
    if (e == hipSuccess) e = hipMalloc(&p, (size_t)capacity * 4);
    if (e == hipSuccess) { P->allocs.push_back(p); P->pos = (int*)p; e = hipMemset(p, 0, (size_t)capacity * 4); }
    if (e != hipSuccess) {
        for (void* q : P->allocs) (void)hipFree(q);
        delete P;
        return fail(c, Q3C_ERR_HIP, std::string("stream pool allocation: ") + hipGetErrorString(e));
    }
    *out = P;
    return 0;
}

void q3c_stream_pool_destroy(q3c_stream_pool* P) {
    if (!P) return;
    for (auto& kv : P->plans)
        for (auto& g : kv.second.graphs) (void)hipGraphExecDestroy(g.exec);
    if (P->capture_stream) (void)hipStreamDestroy(P->capture_stream);
    for (void* q : P->allocs) (void)hipFree(q);
    delete P;
}

int q3c_stream_state_bytes(const q3c_stream_pool* P, size_t* per_slot) {
    if (!P || !per_slot) return Q3C_ERR_INVALID_ARGUMENT;
    *per_slot = (size_t)P->slot_stride;
    return 0;
}

int q3c_stream_reset(q3c_stream_pool* P, const int32_t* slots, int32_t count, void* stream) {
    if (!P) return Q3C_ERR_INVALID_ARGUMENT;
    q3c_ctx* c = P->c;
    if (!slots || count < 1) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_stream_reset: bad argument");
    LAUNCH(c, q3_launch_stream_reset(slots, count, P->capacity, P->state, P->slot_stride, P->pos, c->err_flag, (hipStream_t)stream));
    return 0;
}

int q3c_stream_workspace_bytes(q3c_stream_pool* P, int32_t S, int32_t n, size_t* bytes) {
    if (!P || !bytes) return Q3C_ERR_INVALID_ARGUMENT;
    if (S < 1 || n < 1) return fail(P->c, Q3C_ERR_INVALID_ARGUMENT, "streams and frames must be >= 1");
    StreamPlan* pl;
    TRY(stream_plan(P, S, n, &pl, true));
    *bytes = pl->ws_bytes;
    return 0;
}

int q3c_stream_push(q3c_stream_pool* P, const int32_t* slots, int32_t S, const int64_t* codes, int32_t n, float* wav, void* ws,
                    size_t ws_bytes, void* stream) {
    if (!P) return Q3C_ERR_INVALID_ARGUMENT;
    q3c_ctx* c = P->c;
    if (!slots || !codes || !wav || !ws || S < 1 || n < 1) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_stream_push: bad argument");
    if (S > P->capacity) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "more streams in one push than the pool holds");
    if ((uintptr_t)ws & 255) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
    StreamPlan* pl;
    TRY(stream_plan(P, S, n, &pl, false));
    if (ws_bytes < pl->ws_bytes) return fail(c, Q3C_ERR_WORKSPACE, "workspace too small: need " + std::to_string(pl->ws_bytes));
    return stream_sequence(P, S, n, true, pl, slots, codes, wav, (char*)ws, (hipStream_t)stream, nullptr, nullptr, nullptr);
}

// The same push replayed from a HIP graph.  A push at frame rate is ~90 small launches; replaying them as one graph takes
// the per-launch host work and most of the gaps between dependent kernels out of the latency.  Kernel nodes hold their
// operand addresses, so a graph is keyed by (streams, frames) AND the four buffers; the first call with a new key records
// the sequence on a side stream (nothing runs during recording) and instantiates it, later calls only launch it.
int q3c_stream_push_graph(q3c_stream_pool* P, const int32_t* slots, int32_t S, const int64_t* codes, int32_t n, float* wav, void* ws,
                          size_t ws_bytes, void* stream) {
    if (!P) return Q3C_ERR_INVALID_ARGUMENT;
    q3c_ctx* c = P->c;
    if (!slots || !codes || !wav || !ws || S < 1 || n < 1) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_stream_push_graph: bad argument");
    if (S > P->capacity) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "more streams in one push than the pool holds");
    if ((uintptr_t)ws & 255) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
    StreamPlan* pl;
    TRY(stream_plan(P, S, n, &pl, false));
    if (ws_bytes < pl->ws_bytes) return fail(c, Q3C_ERR_WORKSPACE, "workspace too small: need " + std::to_string(pl->ws_bytes));
    hipStream_t s = (hipStream_t)stream;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (s) HIPCHK(c, hipStreamIsCapturing(s, &st));
    // per-kernel profiling records events between launches, and a caller who is capturing already gets the plain sequence
    if (c->profiling || !c->taps.empty() || st != hipStreamCaptureStatusNone)
        return stream_sequence(P, S, n, true, pl, slots, codes, wav, (char*)ws, s, nullptr, nullptr, nullptr);
    // a graph replays the launch sequence it recorded: besides the four buffers the key holds what chooses that sequence
    for (PushGraph& g : pl->graphs)
        if (g.slots == slots && g.codes == codes && g.wav == wav && g.ws == ws && g.tuning == c->tuning && g.fuse == c->fuse_units) {
            g.used = ++P->graph_clock;
            HIPCHK(c, hipGraphLaunch(g.exec, s));
            return 0;
        }
    // A caller whose workspace moved (it grew) can never present the old address again: those graphs are dead weight.
    for (size_t i = 0; i < pl->graphs.size();)
        if (pl->graphs[i].ws != ws) { (void)hipGraphExecDestroy(pl->graphs[i].exec); pl->graphs.erase(pl->graphs.begin() + (long)i); }
        else ++i;
    if (pl->graphs.size() >= MAX_GRAPHS_PER_PLAN) {     // full: the least recently replayed combination makes room
        size_t lru = 0;
        for (size_t i = 1; i < pl->graphs.size(); ++i) if (pl->graphs[i].used < pl->graphs[lru].used) lru = i;
        (void)hipGraphExecDestroy(pl->graphs[lru].exec);
        pl->graphs.erase(pl->graphs.begin() + (long)lru);
    }
    if (!P->capture_stream) HIPCHK(c, hipStreamCreateWithFlags(&P->capture_stream, hipStreamNonBlocking));
    HIPCHK(c, hipStreamBeginCapture(P->capture_stream, hipStreamCaptureModeThreadLocal));
    const int rc = stream_sequence(P, S, n, true, pl, slots, codes, wav, (char*)ws, P->capture_stream, nullptr, nullptr, nullptr);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(P->capture_stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess || !graph) return fail(c, Q3C_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    PushGraph pg{slots, codes, wav, ws, c->tuning, c->fuse_units, nullptr, ++P->graph_clock};
    const hipError_t e2 = hipGraphInstantiate(&pg.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e2 != hipSuccess) return fail(c, Q3C_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e2));
    pl->graphs.push_back(pg);
    HIPCHK(c, hipGraphLaunch(pg.exec, s));
    return 0;
}

// ---- single operators ---------------------------------------------------------------------------------------
// The operator entry points below build a temporary context per call.  Measurement switches for THOSE contexts only (tools/
// and tests that compare two kernels for one operator); contexts made by q3c_create never read them.
static int op_tuning = 0;
static long long* op_timeline = nullptr;
static int op_timeline_slots = 0;
static void operator_defaults(q3c_ctx* c) { c->tuning = op_tuning; c->timeline = op_timeline; c->timeline_slots = op_timeline_slots; }
int q3c_debug_set_operator_context(int32_t tuning, long long* timeline_dev, int32_t slots) {
    if (slots < 0) return Q3C_ERR_INVALID_ARGUMENT;
    op_tuning = tuning; op_timeline = timeline_dev; op_timeline_slots = timeline_dev ? slots : 0;
    return 0;
}

static int op_run(GemmW& g, q3c_ctx* c, const float* x, int B, int L_in, int cin, float* y, int M, int n_out, int precision,
                  hipStream_t s) {
    const int dt = precision == Q3C_PRECISION_BF16 ? Q3_BF16 : Q3_F32;
    TRY(ensure_zero_page(c));
    void *xi = nullptr, *yo = nullptr;
    int rc = 0;
    if (dt == Q3_BF16) {
        TRY(dev_alloc(c, (size_t)B * L_in * cin * 2, &xi));
        TRY(dev_alloc(c, (size_t)B * M * n_out * 2, &yo));
        LAUNCH(c, q3_launch_cast_from_f32(x, xi, (long long)B * L_in * cin, dt, s));
        rc = gemm(c, g, xi, B, L_in, yo, M, dt, GemmOpts{}, s);
        if (rc) return rc;
        LAUNCH(c, q3_launch_cast_to_f32(yo, y, (long long)B * M * n_out, dt, s));
    } else {
        rc = gemm(c, g, x, B, L_in, y, M, dt, GemmOpts{}, s);
        if (rc) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(s));
    return 0;
}

int q3c_causal_conv1d(const float* x, int32_t B, int32_t L, int32_t cin, int32_t cout, int32_t k, int32_t dil, const float* w,
                      const float* bias, float* y, int32_t precision, void* stream) {
    if (!x || !w || !y || B < 1 || L < 1 || cin % 8 || cout < 1 || k < 1 || dil < 1) return Q3C_ERR_INVALID_ARGUMENT;
    q3c_ctx tmp;
    operator_defaults(&tmp);
    const int dt = precision == Q3C_PRECISION_BF16 ? Q3_BF16 : Q3_F32;
    tmp.staged["w.weight"] = HostTensor{{cout, cin, k}, std::vector<float>(w, w + (size_t)cout * cin * k)};
    if (bias) tmp.staged["w.bias"] = HostTensor{{cout}, std::vector<float>(bias, bias + cout)};
    GemmW g;
    int rc = pack_conv(&tmp, "w", cout, cin, k, dil, dt, &g, bias != nullptr);
    if (!rc) rc = op_run(g, &tmp, x, B, L, cin, y, L, cout, precision, (hipStream_t)stream);
    if (rc) fprintf(stderr, "q3c_causal_conv1d: %s\n", tmp.err.c_str());
    for (void* p : tmp.allocs) (void)hipFree(p);
    return rc;
}

int q3c_causal_tconv1d(const float* x, int32_t B, int32_t L, int32_t cin, int32_t cout, int32_t k, int32_t stride, const float* w,
                       const float* bias, float* y, int32_t precision, void* stream) {
    if (!x || !w || !y || B < 1 || L < 1 || cin < 8 || cin % 8 || cout < 1 || stride < 1 || k < 1) return Q3C_ERR_INVALID_ARGUMENT;
    q3c_ctx tmp;
    operator_defaults(&tmp);
    const int dt = precision == Q3C_PRECISION_BF16 ? Q3_BF16 : Q3_F32;
    GemmW g;
    int rc = pack_tconv_host(&tmp, w, bias, cin, cout, k, stride, dt, &g);
    const int M = (k == stride) ? L : L - 1;
    if (!rc && M < 1) rc = fail(&tmp, Q3C_ERR_INVALID_ARGUMENT, "the two-sided trim leaves no output rows (length 1 with kernel == 2*stride)");
    if (!rc) rc = op_run(g, &tmp, x, B, L, cin, y, M, stride * cout, precision, (hipStream_t)stream);
    if (rc) fprintf(stderr, "q3c_causal_tconv1d: %s\n", tmp.err.c_str());
    for (void* p : tmp.allocs) (void)hipFree(p);
    return rc;
}

int q3c_snake_beta(const float* x, int64_t rows, int32_t C, const float* alpha, const float* beta, float eps, float* y, int32_t precision,
                   void* stream) {
    if (!x || !y || !alpha || !beta || rows < 0 || C < 1) return Q3C_ERR_INVALID_ARGUMENT;
    if (precision != Q3C_PRECISION_F32 && precision != Q3C_PRECISION_BF16) return Q3C_ERR_INVALID_ARGUMENT;
    q3c_ctx tmp;
    operator_defaults(&tmp);
    q3c_ctx* c = &tmp;
    hipStream_t s = (hipStream_t)stream;
    tmp.dtype = precision == Q3C_PRECISION_BF16 ? Q3_BF16 : Q3_F32;       // bf16: hardware sine, frequency in revolutions
    auto run = [&]() -> int {
        Snake sn;
        TRY(pack_snake_host(c, alpha, beta, C, eps, &sn));
        if (tmp.dtype == Q3_F32) {
            LAUNCH(c, q3_launch_snake(x, y, sn.a, sn.invb, rows, C, Q3_F32, s));
        } else {                                                           // bf16 storage on both sides, as inside the decode
            void *xi = nullptr, *yo = nullptr;
            const long long n = (long long)rows * C;
            TRY(dev_alloc(c, (size_t)n * 2, &xi)); TRY(dev_alloc(c, (size_t)n * 2, &yo));
            LAUNCH(c, q3_launch_cast_from_f32(x, xi, n, Q3_BF16, s));
            LAUNCH(c, q3_launch_snake(xi, yo, sn.a, sn.invb, rows, C, Q3_BF16, s));
            LAUNCH(c, q3_launch_cast_to_f32(yo, y, n, Q3_BF16, s));
        }
        HIPCHK(c, hipStreamSynchronize(s));
        return 0;
    };
    const int rc = run();
    for (void* p : tmp.allocs) (void)hipFree(p);
    return rc;
}


// x + conv_k1(snake(conv_k7,dilation(snake(x)))) on x [B][L][C]: the decoder's residual unit, through the same launch the
// decode uses for it (one fused launch when the channel count allows, else two).
int q3c_residual_unit(const float* x, int32_t B, int32_t L, int32_t C, int32_t dilation, const float* alpha1, const float* beta1,
                      const float* w1, const float* b1, const float* alpha2, const float* beta2, const float* w2, const float* b2,
                      float eps, float* y, int32_t precision, void* stream) {
    if (!x || !y || !alpha1 || !beta1 || !w1 || !b1 || !alpha2 || !beta2 || !w2 || !b2 || B < 1 || L < 1 || C < 8 || C % 8 || dilation < 1)
        return Q3C_ERR_INVALID_ARGUMENT;
    q3c_ctx tmp;
    operator_defaults(&tmp);
    q3c_ctx* c = &tmp;
    hipStream_t s = (hipStream_t)stream;
    const int dt = precision == Q3C_PRECISION_BF16 ? Q3_BF16 : Q3_F32;
    tmp.dtype = dt;
    const size_t n = (size_t)B * L * C, es = esize(dt);
    auto run = [&]() -> int {
        tmp.staged["c1.weight"] = HostTensor{{C, C, 7}, std::vector<float>(w1, w1 + (size_t)C * C * 7)};
        tmp.staged["c1.bias"] = HostTensor{{C}, std::vector<float>(b1, b1 + C)};
        tmp.staged["c2.weight"] = HostTensor{{C, C, 1}, std::vector<float>(w2, w2 + (size_t)C * C)};
        tmp.staged["c2.bias"] = HostTensor{{C}, std::vector<float>(b2, b2 + C)};
        GemmW c1, c2;
        Snake a1, a2;
        TRY(pack_conv(c, "c1", C, C, 7, dilation, dt, &c1));
        TRY(pack_conv(c, "c2", C, C, 1, 1, dt, &c2));
        TRY(pack_snake_host(c, alpha1, beta1, C, eps, &a1));
        TRY(pack_snake_host(c, alpha2, beta2, C, eps, &a2));
        TRY(ensure_zero_page(c));
        void *raw = nullptr, *sn = nullptr, *h = nullptr, *out = nullptr;
        TRY(dev_alloc(c, n * es, &raw)); TRY(dev_alloc(c, n * es, &sn)); TRY(dev_alloc(c, n * es, &h)); TRY(dev_alloc(c, n * es, &out));
        LAUNCH(c, q3_launch_cast_from_f32(x, raw, (long long)n, dt, s));
        if (q3_convgemm_fusable(C, dt, c->tuning)) {       // the fused launch reads the raw input and applies the first SnakeBeta itself
            GemmOpts g; g.act = Q3_ACT_SNAKE; g.epi = &a2; g.second = &c2; g.pre = &a1; g.R = raw;
            TRY(gemm(c, c1, raw, B, L, out, L, dt, g, s));
        } else {
            LAUNCH(c, q3_launch_snake(raw, sn, a1.a, a1.invb, (long long)B * L, C, dt, s));
            GemmOpts g1; g1.act = Q3_ACT_SNAKE; g1.epi = &a2;
            TRY(gemm(c, c1, sn, B, L, h, L, dt, g1, s));
            GemmOpts g2; g2.R = raw;
            TRY(gemm(c, c2, h, B, L, out, L, dt, g2, s));
        }
        LAUNCH(c, q3_launch_cast_to_f32(out, y, (long long)n, dt, s));
        HIPCHK(c, hipStreamSynchronize(s));
        return 0;
    };
    const int rc = run();
    if (rc) fprintf(stderr, "q3c_residual_unit: %s\n", tmp.err.c_str());
    for (void* p : tmp.allocs) (void)hipFree(p);
    return rc;
}

// ConvNeXt block on x [B][L][C]: depthwise causal conv k7 -> LayerNorm -> Linear(C,4C) -> GELU -> Linear(4C,C) -> gamma -> + x
int q3c_convnext_block(const float* x, int32_t B, int32_t L, int32_t C, const float* dw_w, const float* dw_b, const float* ln_w,
                       const float* ln_b, const float* pw1_w, const float* pw1_b, const float* pw2_w, const float* pw2_b,
                       const float* gamma, float ln_eps, float* y, int32_t precision, void* stream) {
    if (!x || !y || !dw_w || !dw_b || !ln_w || !ln_b || !pw1_w || !pw1_b || !pw2_w || !pw2_b || !gamma || B < 1 || L < 1 || C < 8 || C % 8)
        return Q3C_ERR_INVALID_ARGUMENT;
    q3c_ctx tmp;
    operator_defaults(&tmp);
    q3c_ctx* c = &tmp;
    hipStream_t s = (hipStream_t)stream;
    const int dt = precision == Q3C_PRECISION_BF16 ? Q3_BF16 : Q3_F32;
    tmp.dtype = dt;
    const size_t n = (size_t)B * L * C, es = esize(dt);
    auto run = [&]() -> int {
        tmp.staged["pw1.w"] = HostTensor{{4 * C, C}, std::vector<float>(pw1_w, pw1_w + (size_t)4 * C * C)};
        tmp.staged["pw1.b"] = HostTensor{{4 * C}, std::vector<float>(pw1_b, pw1_b + 4 * C)};
        tmp.staged["pw2.w"] = HostTensor{{C, 4 * C}, std::vector<float>(pw2_w, pw2_w + (size_t)4 * C * C)};
        tmp.staged["pw2.b"] = HostTensor{{C}, std::vector<float>(pw2_b, pw2_b + C)};
        GemmW pw1, pw2;
        TRY(pack_linear(c, "pw1.w", "pw1.b", 4 * C, C, dt, &pw1));
        TRY(pack_linear(c, "pw2.w", "pw2.b", C, 4 * C, dt, &pw2));
        float *d_dw = nullptr, *d_db = nullptr, *d_lw = nullptr, *d_lb = nullptr, *d_g = nullptr;
        std::vector<float> dwt((size_t)7 * C);                   // [C][1][7] -> tap-major [7][C]
        for (int ch = 0; ch < C; ++ch)
            for (int j = 0; j < 7; ++j) dwt[(size_t)j * C + ch] = dw_w[(size_t)ch * 7 + j];
        TRY(upload_f32(c, dwt, &d_dw));
        TRY(upload_f32(c, std::vector<float>(dw_b, dw_b + C), &d_db));
        TRY(upload_f32(c, std::vector<float>(ln_w, ln_w + C), &d_lw));
        TRY(upload_f32(c, std::vector<float>(ln_b, ln_b + C), &d_lb));
        TRY(upload_f32(c, std::vector<float>(gamma, gamma + C), &d_g));
        TRY(ensure_zero_page(c));
        void *xi = nullptr, *d = nullptr, *pw = nullptr;
        TRY(dev_alloc(c, n * es, &xi)); TRY(dev_alloc(c, n * es, &d)); TRY(dev_alloc(c, 4 * n * es, &pw));
        LAUNCH(c, q3_launch_cast_from_f32(x, xi, (long long)n, dt, s));
        LAUNCH(c, q3_launch_dwconv_ln(xi, d_dw, d_db, d_lw, d_lb, d, B, L, C, 7, ln_eps, dt, s));
        GemmOpts g1; g1.act = Q3_ACT_GELU;
        TRY(gemm(c, pw1, d, B, L, pw, L, dt, g1, s));
        GemmOpts g2; g2.scale = d_g; g2.R = xi;
        TRY(gemm(c, pw2, pw, B, L, xi, L, dt, g2, s));
        LAUNCH(c, q3_launch_cast_to_f32(xi, y, (long long)n, dt, s));
        HIPCHK(c, hipStreamSynchronize(s));
        return 0;
    };
    const int rc = run();
    if (rc) fprintf(stderr, "q3c_convnext_block: %s\n", tmp.err.c_str());
    for (void* p : tmp.allocs) (void)hipFree(p);
    return rc;
}

// The context's transformer stack alone (in_proj, sliding-window layers, final norm, out_proj) on hidden states
// x [B][T][latent_dim] in fp32; positions start at 0 as in a chunk of the decode.
int q3c_transformer(q3c_ctx* c, const float* x, int32_t B, int32_t T, float* y, void* ws, size_t ws_bytes, void* stream) {
    if (!c) return Q3C_ERR_INVALID_ARGUMENT;
    if (!c->finalized) return fail(c, Q3C_ERR_STATE, "q3c_finalize has not run");
    if (!x || !y || !ws || B < 1 || T < 1) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "q3c_transformer: bad argument");
    if (T > c->cfg.max_positions) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "frames exceed max_positions");
    if ((uintptr_t)ws & 255) return fail(c, Q3C_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
    const Layout lay = make_layout(c, B, T);
    if (ws_bytes < lay.total) return fail(c, Q3C_ERR_WORKSPACE, "workspace too small: need " + std::to_string(lay.total));
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)ws;
    const long long n = (long long)B * T * c->cfg.latent_dim;
    LAUNCH(c, q3_launch_cast_from_f32(x, base + lay.oA, n, c->dtype, s));
    void *cur, *other;
    TRY(transformer_stack(c, B, T, base + lay.oA, base + lay.oB, base + lay.oC, s, &cur, &other));
    LAUNCH(c, q3_launch_cast_to_f32(cur, y, n, c->dtype, s));
    return 0;
}

// fp32 samples -> 16-bit PCM on the device (asynchronous on the stream)
int q3c_wav_to_pcm16(const float* wav, int64_t samples, int16_t* pcm, void* stream) {
    if ((!wav || !pcm) && samples > 0) return Q3C_ERR_INVALID_ARGUMENT;
    if (samples < 0) return Q3C_ERR_INVALID_ARGUMENT;
    return q3_launch_pcm16(wav, (short*)pcm, (long long)samples, (hipStream_t)stream) == hipSuccess ? 0 : Q3C_ERR_HIP;
}

}  // extern "C"

#include "encoder_impl.h"
