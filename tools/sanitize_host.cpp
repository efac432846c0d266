// Host-side sanitizer driver (built and run by tools/sanitize_host.py with -fsanitize=address,undefined on the host pass and
// -DQ3C_HOST_DRYRUN, where device memory is host memory): walks the library's host logic - configuration checks, weight
// staging, q3c_finalize's repacking, workspace and stream-state layouts, every error path that needs no launch - on weight
// blobs written by the Python side.  Blob: int32 count, then per tensor: int32 name length, name, int32 ndim, int64 shape[ndim],
// float data.
#include "../include/q3c.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct Tensor { std::string name; std::vector<int64_t> shape; std::vector<float> data; };

static std::vector<Tensor> read_blob(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
    auto rd = [&](void* p, size_t n) { if (std::fread(p, 1, n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); } };
    int32_t count; rd(&count, 4);
    std::vector<Tensor> out((size_t)count);
    for (auto& t : out) {
        int32_t nl, nd; rd(&nl, 4); t.name.resize((size_t)nl); rd(&t.name[0], (size_t)nl); rd(&nd, 4);
        t.shape.resize((size_t)nd); rd(t.shape.data(), 8 * (size_t)nd);
        size_t n = 1; for (auto d : t.shape) n *= (size_t)d;
        t.data.resize(n); rd(t.data.data(), 4 * n);
    }
    std::fclose(f);
    return out;
}

static int fails = 0;
#define EXPECT(cond) do { if (!(cond)) { std::fprintf(stderr, "%s:%d: EXPECT(%s) failed\n", __FILE__, __LINE__, #cond); ++fails; } } while (0)

static void load_all(q3c_ctx* c, const std::vector<Tensor>& w, const char* skip = nullptr) {
    for (const auto& t : w)
        if (!skip || t.name != skip) EXPECT(q3c_load_weight(c, t.name.c_str(), t.data.data(), t.shape.data(), (int32_t)t.shape.size()) == 0);
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: sanitize_host config.bin weights.bin [encoder_config.bin]\n"); return 2; }
    q3c_config cfg;
    { FILE* f = std::fopen(argv[1], "rb"); if (!f || std::fread(&cfg, 1, sizeof(cfg), f) != sizeof(cfg)) { std::fprintf(stderr, "bad config file\n"); return 2; } std::fclose(f); }
    const std::vector<Tensor> w = read_blob(argv[2]);
    q3c_encoder_config ecfg{};
    const bool enc = argc > 3;
    if (enc) { FILE* f = std::fopen(argv[3], "rb"); if (!f || std::fread(&ecfg, 1, sizeof(ecfg), f) != sizeof(ecfg)) { std::fprintf(stderr, "bad encoder config\n"); return 2; } std::fclose(f); }

    // ---- argument and configuration checks -----------------------------------------------------------------------------
    q3c_ctx* c = nullptr;
    EXPECT(q3c_create(nullptr, &c) == Q3C_ERR_INVALID_ARGUMENT);
    EXPECT(q3c_create(&cfg, nullptr) == Q3C_ERR_INVALID_ARGUMENT);
    {
        q3c_config bad = cfg; bad.struct_size = 12;
        EXPECT(q3c_create(&bad, &c) == Q3C_ERR_INVALID_ARGUMENT && c && std::strlen(q3c_last_error(c)) > 0); q3c_destroy(c);
        bad = cfg; bad.num_upsample_rates = Q3C_MAX_STAGES + 1;
        EXPECT(q3c_create(&bad, &c) == Q3C_ERR_INVALID_ARGUMENT); q3c_destroy(c);
        bad = cfg; bad.num_dilations = -1;
        EXPECT(q3c_create(&bad, &c) == Q3C_ERR_INVALID_ARGUMENT); q3c_destroy(c);
        bad = cfg; bad.sliding_window = 1000;
        EXPECT(q3c_create(&bad, &c) == Q3C_ERR_UNSUPPORTED); q3c_destroy(c);
        bad = cfg; bad.head_dim = 7;
        EXPECT(q3c_create(&bad, &c) != 0); q3c_destroy(c);
        bad = cfg; bad.precision = 9;
        EXPECT(q3c_create(&bad, &c) == Q3C_ERR_INVALID_ARGUMENT); q3c_destroy(c);
    }
    q3c_destroy(nullptr);
    EXPECT(std::strlen(q3c_last_error(nullptr)) > 0 && std::strlen(q3c_version()) > 0);

    // ---- a missing tensor, a mis-shaped tensor, call-order errors -----------------------------------------------------------
    EXPECT(q3c_create(&cfg, &c) == 0);
    size_t bytes = 0;
    EXPECT(q3c_finalize(c) == Q3C_ERR_WEIGHT);                       // nothing loaded
    q3c_destroy(c);
    EXPECT(q3c_create(&cfg, &c) == 0);
    load_all(c, w, w.front().name.c_str());
    EXPECT(q3c_finalize(c) == Q3C_ERR_WEIGHT && std::strstr(q3c_last_error(c), w.front().name.c_str()));
    q3c_destroy(c);
    EXPECT(q3c_create(&cfg, &c) == 0);
    load_all(c, w);
    {
        const Tensor& t = w[w.size() / 2];
        std::vector<int64_t> sh = t.shape; sh[0] += 1;
        std::vector<float> d(t.data.size() / (size_t)t.shape[0] * (size_t)sh[0]);
        EXPECT(q3c_load_weight(c, t.name.c_str(), d.data(), sh.data(), (int32_t)sh.size()) == 0);       // replaces the staged tensor
        EXPECT(q3c_finalize(c) == Q3C_ERR_WEIGHT && std::strstr(q3c_last_error(c), "shape"));
        int64_t neg[1] = {-1};
        EXPECT(q3c_load_weight(c, "x", d.data(), neg, 1) == Q3C_ERR_WEIGHT);
        EXPECT(q3c_load_weight(c, nullptr, d.data(), sh.data(), 1) == Q3C_ERR_INVALID_ARGUMENT);
        EXPECT(q3c_load_weight(c, "x", d.data(), sh.data(), 9) == Q3C_ERR_INVALID_ARGUMENT);
    }
    float dummy[4] = {0};
    int64_t codes[4] = {0};
    EXPECT(q3c_forward(c, codes, 1, 1, dummy, dummy, 16, nullptr) == Q3C_ERR_STATE);                     // before finalize
    q3c_stream_pool* pool = nullptr;
    EXPECT(q3c_stream_pool_create(c, 4, &pool) == Q3C_ERR_STATE);
    q3c_destroy(c);

    // ---- the full load: every repacking loop, the layouts, the stream plans ----------------------------------------------------
    EXPECT(q3c_create(&cfg, &c) == 0);
    if (enc) {
        q3c_encoder_config bad = ecfg; bad.struct_size = 3;
        EXPECT(q3c_encoder_configure(c, &bad) != 0);
        EXPECT(q3c_encoder_configure(c, &ecfg) == 0);
    }
    load_all(c, w);
    EXPECT(q3c_finalize(c) == 0);
    if (fails) { std::fprintf(stderr, "finalize: %s\n", q3c_last_error(c)); }
    EXPECT(q3c_finalize(c) == Q3C_ERR_STATE);
    EXPECT(q3c_load_weight(c, w[0].name.c_str(), w[0].data.data(), w[0].shape.data(), (int32_t)w[0].shape.size()) == Q3C_ERR_STATE);
    int64_t samples = 0;
    EXPECT(q3c_output_length(c, 125, &samples) == 0 && samples > 0);
    EXPECT(q3c_output_length(c, 0, &samples) == Q3C_ERR_INVALID_ARGUMENT);
    const int shapes[][2] = {{1, 1}, {1, 2}, {3, 73}, {64, 300}, {1, 7500 < cfg.max_positions ? 7500 : cfg.max_positions}};
    size_t prev = 0;
    for (const auto& s : shapes) {
        EXPECT(q3c_workspace_bytes(c, s[0], s[1], &bytes) == 0 && bytes > 0 && bytes % 256 == 0);
        if (s[0] * s[1] > 1) EXPECT(bytes >= prev);
        prev = s[0] * s[1] == 1 ? bytes : prev;
    }
    EXPECT(q3c_workspace_bytes(c, 0, 5, &bytes) == Q3C_ERR_INVALID_ARGUMENT);
    EXPECT(q3c_forward(c, nullptr, 1, 1, dummy, dummy, 16, nullptr) == Q3C_ERR_INVALID_ARGUMENT);
    EXPECT(q3c_forward(c, codes, 1, 1, dummy, (void*)((char*)dummy + 4), 1 << 20, nullptr) == Q3C_ERR_INVALID_ARGUMENT);      // misaligned workspace
    alignas(256) static unsigned char small_ws[256];
    EXPECT(q3c_forward(c, codes, 1, 1, dummy, small_ws, sizeof(small_ws), nullptr) == Q3C_ERR_WORKSPACE);
    EXPECT(q3c_forward(c, codes, 1, cfg.max_positions + 1, dummy, small_ws, sizeof(small_ws), nullptr) == Q3C_ERR_INVALID_ARGUMENT);
    EXPECT(q3c_forward_skip(c, codes, 1, 4, 4, dummy, small_ws, sizeof(small_ws), nullptr) == Q3C_ERR_INVALID_ARGUMENT);
    EXPECT(q3c_forward_scatter(c, codes, 1, 4, 0, nullptr, dummy, small_ws, sizeof(small_ws), nullptr) == Q3C_ERR_INVALID_ARGUMENT);
    EXPECT(q3c_forward_static(c, codes, 4096, 4096 < cfg.max_positions ? 4096 : cfg.max_positions, dummy, nullptr) == Q3C_ERR_WORKSPACE);
    EXPECT(q3c_set_tap(c, Q3C_STAGE_CONV_IN, dummy, 4) == 0 && q3c_set_tap(c, Q3C_STAGE_CONV_IN, nullptr, 0) == 0);
    EXPECT(q3c_debug_set_tuning(c, 64) == 0 && q3c_debug_set_tuning(c, 0) == 0 && q3c_debug_set_tuning(nullptr, 0) != 0);
    EXPECT(q3c_profile_enable(c, 1) == 0 && q3c_profile_enable(c, 0) == 0);
    q3c_profile_entry pe[4]; int32_t npe = -1;
    EXPECT(q3c_profile_read(c, pe, 4, &npe) == 0 && npe == 0);
    if (enc) {
        int64_t fr = 0;
        EXPECT(q3c_encode_frames(c, 98765, &fr) == 0 && fr > 0);
        EXPECT(q3c_encode_workspace_bytes(c, 2, 98765, &bytes) == 0 && bytes > 0);
        EXPECT(q3c_encode_workspace_bytes(c, 0, 98765, &bytes) != 0);
    }
    EXPECT(q3c_stream_pool_create(c, 0, &pool) == Q3C_ERR_INVALID_ARGUMENT);
    EXPECT(q3c_stream_pool_create(c, 6, &pool) == 0 && pool);
    size_t per_slot = 0;
    EXPECT(q3c_stream_state_bytes(pool, &per_slot) == 0 && per_slot > 0 && per_slot % 256 == 0);
    const int pushes[][2] = {{1, 1}, {6, 1}, {3, 5}, {2, 25}, {6, 1}};
    for (const auto& p : pushes) EXPECT(q3c_stream_workspace_bytes(pool, p[0], p[1], &bytes) == 0 && bytes > 0);
    EXPECT(q3c_stream_workspace_bytes(pool, 0, 1, &bytes) == Q3C_ERR_INVALID_ARGUMENT);
    int32_t slots[2] = {0, 1};
    EXPECT(q3c_stream_push(pool, slots, 7, codes, 1, dummy, small_ws, sizeof(small_ws), nullptr) == Q3C_ERR_INVALID_ARGUMENT);      // more streams than slots
    EXPECT(q3c_stream_push(pool, slots, 2, codes, 9, dummy, small_ws, sizeof(small_ws), nullptr) == Q3C_ERR_STATE);                 // pair never prepared
    EXPECT(q3c_stream_push(pool, slots, 6, codes, 1, dummy, small_ws, sizeof(small_ws), nullptr) == Q3C_ERR_WORKSPACE);
    EXPECT(q3c_stream_reset(pool, nullptr, 1, nullptr) == Q3C_ERR_INVALID_ARGUMENT);
    q3c_stream_pool_destroy(pool);
    q3c_stream_pool_destroy(nullptr);
    q3c_destroy(c);
    std::printf("sanitize_host: %zu tensors, state %zu bytes per stream, %d failed expectation(s)\n", w.size(), per_slot, fails);
    return fails ? 1 : 0;
}
