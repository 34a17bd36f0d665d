// api.hip — the C ABI of libgapfill_hip.so (include/gapfill_hip.h): context, staging, host/device variants.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "gf_internal.hpp"

namespace gf {

int set_hip_error(gf_ctx* ctx, hipError_t e, const char* what) {
    if (ctx) ctx->last_error = std::string(hipGetErrorString(e)) + " in " + what;
    return e == hipErrorOutOfMemory ? GF_E_NOMEM : GF_E_NODEV;
}

int ensure(gf_ctx* ctx, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return GF_OK;
    if (b.p) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(b.p);
        b.p = nullptr;
        b.bytes = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    GF_HIP(ctx, hipMalloc(&b.p, want));
    b.bytes = want;
    return GF_OK;
}

__global__ void zero_regions_kernel(ZeroList z) {
    for (int r = 0; r < 4; ++r)
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < z.n[r]; i += gridDim.x * blockDim.x) z.p[r][i] = 0;
}

void zero_regions(gf_ctx* ctx, const ZeroList& z) {
    uint32_t most = 0;
    for (int r = 0; r < 4; ++r) most = std::max(most, z.n[r]);
    if (!most) return;
    hipLaunchKernelGGL(zero_regions_kernel, dim3(std::min<uint32_t>((most + 255) / 256, 1024)), dim3(256), 0, ctx->stream, z);
}

static void drain_timing(gf_ctx* ctx) {
    for (auto& l : ctx->launches) {
        float ms = 0;
        if (hipEventSynchronize(l.b) == hipSuccess && hipEventElapsedTime(&ms, l.a, l.b) == hipSuccess) {
            ctx->t_total[l.which] += ms;
            ctx->t_count[l.which] += 1;
        }
        (void)hipEventDestroy(l.a);
        (void)hipEventDestroy(l.b);
    }
    ctx->launches.clear();
}

}  // namespace gf

using namespace gf;

static bool taghit_less(const gf_taghit& a, const gf_taghit& b) {
    if (a.rec != b.rec) return a.rec < b.rec;
    if (a.gap != b.gap) return a.gap < b.gap;
    return a.kind < b.kind;
}

// recs on the host (staged to HBM here) or, with d_src, already on the device (the records gf_bam_pack left there)
template <typename F>
static int tag_host(gf_ctx* ctx, const gf_alnrec* recs, size_t n, gf_taghit* out, size_t cap, size_t* n_out, F launch,
                    const void* d_src = nullptr) {
    if (!ctx || !n_out || (n && !recs && !d_src) || (cap && !out)) return GF_E_INVAL;
    *n_out = 0;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if (!d_src && (rc = ensure(ctx, ctx->stage_in, n * sizeof(gf_alnrec) + 64))) return rc;
    if ((rc = ensure(ctx, ctx->stage_out, cap * sizeof(gf_taghit) + 64))) return rc;
    uint32_t* d_n = (uint32_t*)((uint8_t*)ctx->stage_out.p + cap * sizeof(gf_taghit));
    d_n = (uint32_t*)(((uintptr_t)d_n + 15) & ~(uintptr_t)15);
    if (n && !d_src) GF_HIP(ctx, hipMemcpyAsync(ctx->stage_in.p, recs, n * sizeof(gf_alnrec), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch(d_src ? const_cast<void*>(d_src) : ctx->stage_in.p, ctx->stage_out.p, d_n))) return rc;
    uint32_t cnt = 0;
    GF_HIP(ctx, hipMemcpyAsync(&cnt, d_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = cnt;
    if (cnt > cap) return GF_E_NOSPACE;
    if (cnt) GF_HIP(ctx, hipMemcpy(out, ctx->stage_out.p, (size_t)cnt * sizeof(gf_taghit), hipMemcpyDeviceToHost));
    std::sort(out, out + cnt, taghit_less);
    return GF_OK;
}


extern "C" {

const char* gf_strerror(int code) {
    switch (code) {
        case GF_OK: return "ok";
        case GF_E_INVAL: return "invalid argument";
        case GF_E_NODEV: return "HIP device/runtime error";
        case GF_E_NOMEM: return "out of memory";
        case GF_E_NOSPACE: return "output capacity too small";
        case GF_E_STATE: return "call order violated";
        case GF_E_UNSUPPORTED: return "unsupported parameter";
        case GF_E_FORMAT: return "malformed input bytes";
        default: return "unknown error";
    }
}

// live contexts: gf_destroy disarms every gf_stream_wait_after_filter that still points at the context going away
static std::mutex g_live_mu;
static std::vector<gf_ctx*> g_live;

int gf_init(int device_ordinal, gf_ctx** out) {
    if (!out) return GF_E_INVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_ordinal < 0 || device_ordinal >= n) return GF_E_NODEV;
    gf_ctx* ctx = new (std::nothrow) gf_ctx();
    if (!ctx) return GF_E_NOMEM;
    ctx->device = device_ordinal;
    if (hipSetDevice(device_ordinal) != hipSuccess || hipStreamCreate(&ctx->own_stream) != hipSuccess) {
        delete ctx;
        return GF_E_NODEV;
    }
    ctx->stream = ctx->own_stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0)
        ctx->n_cu = prop.multiProcessorCount;
    if (const char* e = getenv("GF_BITMAP_LOG2")) ctx->bitmap_log2_override = atoi(e);
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live.push_back(ctx);
    }
    *out = ctx;
    return GF_OK;
}

void gf_destroy(gf_ctx* ctx) {
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live.erase(std::remove(g_live.begin(), g_live.end(), ctx), g_live.end());
        for (gf_ctx* other : g_live)
            if (other->after_filter == ctx) other->after_filter = nullptr;
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    drain_timing(ctx);
    for (auto& kv : ctx->index) free_flank_index(ctx, kv.second);
    for (DevBuf* b : {&ctx->cand, &ctx->cand2, &ctx->part_ws, &ctx->tag_stage, &ctx->verify_stage, &ctx->bam_stream, &ctx->bam_recs, &ctx->asm_table, &ctx->asm_surv, &ctx->asm_nodes, &ctx->asm_jump, &ctx->asm_big, &ctx->rowgap, &ctx->pool_ws, &ctx->xchg_ws, &ctx->xchg_ws2, &ctx->counters, &ctx->stage_in, &ctx->stage_out, &ctx->stage_aux, &ctx->table})
        if (b->p) (void)hipFree(b->p);
    for (auto& kv : ctx->anchor_tabs) if (kv.second.p) (void)hipFree(kv.second.p);
    drop_tag_maps(ctx);
    if (ctx->d_gaps) (void)hipFree(ctx->d_gaps);
    if (ctx->d_scaf_off) (void)hipFree(ctx->d_scaf_off);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char* gf_last_error(gf_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }
const char* gf_screen_kernels(gf_ctx* ctx) { return ctx ? ctx->screen_kernels.c_str() : ""; }

int gf_set_stream(gf_ctx* ctx, void* s) {
    if (!ctx) return GF_E_INVAL;
    ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
    return GF_OK;
}

int gf_sync(gf_ctx* ctx) {
    if (!ctx) return GF_E_INVAL;
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GF_OK;
}

int gf_stream_wait(gf_ctx* waiter, gf_ctx* producer) {
    if (!waiter || !producer) return GF_E_INVAL;
    if (waiter->device != producer->device) return GF_E_INVAL;
    GF_HIP(waiter, hipSetDevice(waiter->device));
    hipEvent_t ev;
    GF_HIP(waiter, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, producer->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(waiter->stream, ev, 0);
    (void)hipEventDestroy(ev);   // destruction is deferred until the event has completed
    if (e != hipSuccess) return gf::set_hip_error(waiter, e, "gf_stream_wait");
    return GF_OK;
}

int gf_stream_wait_after_filter(gf_ctx* waiter, gf_ctx* producer) {
    if (!waiter || !producer || waiter == producer || waiter->device != producer->device) return GF_E_INVAL;
    producer->after_filter = waiter;
    return GF_OK;
}

int gf_set_option(gf_ctx* ctx, const char* name, long value) {
    if (!ctx || !name) return GF_E_INVAL;
    if (!strcmp(name, "max_gaps_per_kmer")) { ctx->max_gaps_per_kmer = value < 0 ? 0 : (uint32_t)value; return GF_OK; }
    if (!strcmp(name, "bitmap_log2")) {
        ctx->bitmap_log2_override = (int)value;
        for (auto& kv : ctx->index) free_flank_index(ctx, kv.second);
        ctx->index.clear();
        return GF_OK;
    }
    if (!strcmp(name, "index_host")) {   // test comparator (host-built index), never chosen automatically; takes effect for
        if (value && !getenv("GF_DIAGNOSTICS")) return GF_E_UNSUPPORTED;   // indexes built afterwards
        ctx->index_host = value != 0;
        for (auto& kv : ctx->index) free_flank_index(ctx, kv.second);
        ctx->index.clear();
        return GF_OK;
    }
    if (!strcmp(name, "screen_variant")) { ctx->screen_variant = (int)value; return GF_OK; }
    if (!strcmp(name, "asm_dbg_ptr")) { ctx->asm_dbg = (void*)(uintptr_t)value; return GF_OK; }
    if (!strcmp(name, "asm_stats_ptr")) { ctx->asm_stats = (void*)(uintptr_t)value; return GF_OK; }
    if (!strcmp(name, "tag_dbg")) { if (!getenv("GF_DIAGNOSTICS")) return GF_E_INVAL; ctx->tag_dbg = (int)value; return GF_OK; }
    if (!strcmp(name, "tag_light")) { ctx->tag_light = value != 0; return GF_OK; }
    if (!strcmp(name, "asm_tiebreak")) { if (value < 0 || value > 1) return GF_E_INVAL; ctx->asm_tiebreak = (int)value; return GF_OK; }
    if (!strcmp(name, "asm_keyslot")) { ctx->asm_keyslot = value != 0; return GF_OK; }
    if (!strcmp(name, "asm_sweep")) { if (value < 0 || value > 1) return GF_E_INVAL; ctx->asm_sweep = (int)value; return GF_OK; }
    if (!strcmp(name, "asm_pre_frac8")) { if (value < 1 || value > 7) return GF_E_INVAL; ctx->asm_pre_frac8 = (int)value; return GF_OK; }
    if (!strcmp(name, "asm_precount")) { ctx->asm_precount = value != 0; return GF_OK; }
    if (!strcmp(name, "asm_ranked")) { ctx->asm_ranked = value != 0; return GF_OK; }
    if (!strcmp(name, "asm_max_pool_reads")) { if (value < 0 || value > 0x3FFFFFFF) return GF_E_INVAL; ctx->asm_max_pool_reads = value; return GF_OK; }
    if (!strcmp(name, "asm_big_pool_reads")) { if (value < 0 || value > 0x1FFFFF) return GF_E_INVAL; ctx->asm_big_pool_reads = value; return GF_OK; }
    if (!strcmp(name, "asm_simplify")) { if (value < 0 || value > 8) return GF_E_INVAL; ctx->asm_simplify = (int)value; return GF_OK; }
    if (!strcmp(name, "asm_lds_pool_kb")) { ctx->asm_lds_pool_kb = (int)value; return GF_OK; }
    if (!strcmp(name, "asm_threads")) { if (value != 0 && value != 1024 && value != 512 && value != 256) return GF_E_INVAL; ctx->asm_threads = (int)value; return GF_OK; }
    if (!strcmp(name, "screen_lds_log2_max")) {
        ctx->screen_lds_log2_max = std::max(15, std::min(20, (int)value));
        for (auto& kv : ctx->index) free_flank_index(ctx, kv.second);
        ctx->index.clear();
        return GF_OK;
    }
    if (!strcmp(name, "tag_bins_log2")) { ctx->tag_bins_log2 = std::max(13, std::min(19, (int)value)); drop_tag_maps(ctx); return GF_OK; }
    if (!strcmp(name, "tag_fine_log2")) { ctx->tag_fine_log2 = std::max(20, std::min(28, (int)value)); drop_tag_maps(ctx); return GF_OK; }
    if (!strcmp(name, "tag_nt")) { ctx->tag_nt = value != 0; return GF_OK; }
    if (!strcmp(name, "screen_pf4_cap8")) { ctx->screen_pf4_cap8 = (int)value; return GF_OK; }
    if (!strcmp(name, "screen_ext")) { ctx->screen_ext = value != 0; return GF_OK; }
    if (!strcmp(name, "screen_verify_batch")) { ctx->screen_verify_batch = (int)value; return GF_OK; }
    if (!strcmp(name, "screen_verify_ext")) { ctx->screen_verify_ext = value != 0; return GF_OK; }
    if (!strcmp(name, "screen_verify_gate")) { ctx->screen_verify_gate = value != 0; return GF_OK; }
    if (!strcmp(name, "screen_stream_policy")) { ctx->screen_stream_policy = (int)value; return GF_OK; }
    return GF_E_INVAL;
}

int gf_set_gaps(gf_ctx* ctx, const gf_gap* gaps, size_t n_gaps, uint32_t n_scaffolds, const char* flank_ascii,
                const uint64_t* flank_off) {
    if (!ctx || (n_gaps && !gaps) || n_scaffolds == 0) return GF_E_INVAL;
    for (size_t g = 0; g < n_gaps; ++g) {
        if (gaps[g].scaffold >= n_scaffolds || gaps[g].end < gaps[g].start) return GF_E_INVAL;
        if (g && (gaps[g].scaffold < gaps[g - 1].scaffold ||
                  (gaps[g].scaffold == gaps[g - 1].scaffold && gaps[g].start < gaps[g - 1].end)))
            return GF_E_INVAL;  // grouped by scaffold, ascending, non-overlapping — as the reference's scan writes them
    }
    GF_HIP(ctx, hipSetDevice(ctx->device));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& kv : ctx->index) free_flank_index(ctx, kv.second);
    ctx->index.clear();
    ctx->gaps.assign(gaps, gaps + n_gaps);
    drop_tag_maps(ctx);
    ctx->low_rows.clear();
    ctx->rowgap_rows.clear();
    ctx->n_scaffolds = n_scaffolds;
    ctx->flank_left.assign(n_gaps, std::string());
    ctx->flank_right.assign(n_gaps, std::string());
    if (flank_ascii && flank_off) {
        for (size_t g = 0; g < n_gaps; ++g) {
            if (flank_off[2 * g] > flank_off[2 * g + 1] || flank_off[2 * g + 1] > flank_off[2 * g + 2]) return GF_E_INVAL;
            ctx->flank_left[g].assign(flank_ascii + flank_off[2 * g], flank_ascii + flank_off[2 * g + 1]);
            ctx->flank_right[g].assign(flank_ascii + flank_off[2 * g + 1], flank_ascii + flank_off[2 * g + 2]);
        }
    }
    if (ctx->d_gaps) { (void)hipFree(ctx->d_gaps); ctx->d_gaps = nullptr; }
    if (ctx->d_scaf_off) { (void)hipFree(ctx->d_scaf_off); ctx->d_scaf_off = nullptr; }
    for (auto& kv : ctx->anchor_tabs) if (kv.second.p) (void)hipFree(kv.second.p);
    ctx->anchor_tabs.clear();
    std::vector<uint32_t> off(n_scaffolds + 1, 0);
    for (size_t g = 0; g < n_gaps; ++g) off[gaps[g].scaffold + 1]++;
    for (uint32_t s = 0; s < n_scaffolds; ++s) off[s + 1] += off[s];
    GF_HIP(ctx, hipMalloc((void**)&ctx->d_gaps, std::max<size_t>(1, n_gaps) * sizeof(gf_gap)));
    GF_HIP(ctx, hipMalloc((void**)&ctx->d_scaf_off, off.size() * 4));
    if (n_gaps) GF_HIP(ctx, hipMemcpy(ctx->d_gaps, gaps, n_gaps * sizeof(gf_gap), hipMemcpyHostToDevice));
    GF_HIP(ctx, hipMemcpy(ctx->d_scaf_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    return GF_OK;
}

size_t gf_packed_read_bytes(int read_len) { return read_len <= 0 ? 0 : (size_t)(read_len + 3) / 4; }

int gf_pack_reads(const char* ascii, size_t n_reads, int read_len, uint8_t* packed, uint32_t* n_mask) {
    if (read_len <= 0 || (n_reads && (!ascii || !packed))) return GF_E_INVAL;
    const size_t rb = gf_packed_read_bytes(read_len), nmw = (size_t)(read_len + 31) / 32;
    for (size_t r = 0; r < n_reads; ++r) {
        const char* s = ascii + r * (size_t)read_len;
        uint8_t* o = packed + r * rb;
        memset(o, 0, rb);
        if (n_mask) memset(n_mask + r * nmw, 0, nmw * 4);
        for (int i = 0; i < read_len; ++i) {
            const char c = s[i];
            o[i >> 2] |= (uint8_t)(base_code(c) << (6 - 2 * (i & 3)));
            const bool acgt = c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'a' || c == 'c' || c == 'g' || c == 't';
            if (n_mask && !acgt) n_mask[r * nmw + (i >> 5)] |= 1u << (i & 31);
        }
    }
    return GF_OK;
}

// ---- screen ------------------------------------------------------------------------------------------
int gf_screen_reads_dev(gf_ctx* ctx, const void* d_reads, const void* d_nmask, size_t n_reads, int read_len, int k,
                        int min_hits, void* d_out, size_t cap, void* d_n_out) {
    if (!ctx || !d_n_out || (n_reads && !d_reads) || (cap && !d_out)) return GF_E_INVAL;
    if (ctx->gaps.empty() && ctx->n_scaffolds == 0) return GF_E_STATE;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    FlankIndex* ix = nullptr;
    int rc = build_flank_index(ctx, k, &ix);
    if (rc) return rc;
    return launch_screen(ctx, *ix, d_reads, d_nmask, n_reads, read_len, min_hits, d_out, cap, d_n_out);
}

int gf_screen_last_overflow(gf_ctx* ctx, size_t* n_reads_dropped) {
    if (!ctx || !n_reads_dropped) return GF_E_INVAL;
    *n_reads_dropped = 0;
    if (!ctx->counters.p) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t v = 0;
    GF_HIP(ctx, hipMemcpyAsync(&v, (uint32_t*)ctx->counters.p + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_reads_dropped = v;
    return GF_OK;
}

int gf_screen_reads(gf_ctx* ctx, const uint8_t* packed, const uint32_t* n_mask, size_t n_reads, int read_len, int k,
                    int min_hits, gf_hit* out, size_t cap, size_t* n_out) {
    if (!ctx || !n_out || (n_reads && !packed) || (cap && !out) || read_len <= 0) return GF_E_INVAL;
    *n_out = 0;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t rb = gf_packed_read_bytes(read_len), nmw = (size_t)(read_len + 31) / 32;
    int rc;
    if ((rc = ensure(ctx, ctx->stage_in, n_reads * rb + 64))) return rc;
    if ((rc = ensure(ctx, ctx->stage_out, cap * sizeof(gf_hit) + 64))) return rc;
    if (n_mask && (rc = ensure(ctx, ctx->stage_aux, n_reads * nmw * 4 + 64))) return rc;
    uint32_t* d_n = (uint32_t*)((uint8_t*)ctx->stage_out.p + cap * sizeof(gf_hit));
    d_n = (uint32_t*)(((uintptr_t)d_n + 15) & ~(uintptr_t)15);
    if (n_reads) GF_HIP(ctx, hipMemcpyAsync(ctx->stage_in.p, packed, n_reads * rb, hipMemcpyHostToDevice, ctx->stream));
    if (n_mask && n_reads)
        GF_HIP(ctx, hipMemcpyAsync(ctx->stage_aux.p, n_mask, n_reads * nmw * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = gf_screen_reads_dev(ctx, ctx->stage_in.p, n_mask ? ctx->stage_aux.p : nullptr, n_reads, read_len, k, min_hits,
                             ctx->stage_out.p, cap, d_n);
    if (rc) return rc;
    uint32_t cnt[2] = {0, 0};
    GF_HIP(ctx, hipMemcpyAsync(&cnt[0], d_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(&cnt[1], (uint32_t*)ctx->counters.p + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (cnt[1]) return GF_E_UNSUPPORTED;  // a read matched more (position, gap) pairs than the verify list holds
    *n_out = cnt[0];
    if (cnt[0] > cap) return GF_E_NOSPACE;
    if (cnt[0]) GF_HIP(ctx, hipMemcpy(out, ctx->stage_out.p, (size_t)cnt[0] * sizeof(gf_hit), hipMemcpyDeviceToHost));
    std::sort(out, out + cnt[0], [](const gf_hit& a, const gf_hit& b) { return a.gap < b.gap || (a.gap == b.gap && a.read < b.read); });
    return GF_OK;
}

// ---- tagger --------------------------------------------------------------------------------------------
int gf_tag_alignments_dev(gf_ctx* ctx, const void* d_recs, size_t n, int insert_size, int sd, int clip_dist,
                          int anchor_mapq, void* d_out, size_t cap, void* d_n_out) {
    if (!ctx || !d_n_out || (n && !d_recs) || (cap && !d_out)) return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    return launch_tag(ctx, d_recs, n, insert_size, sd, clip_dist, anchor_mapq, d_out, cap, d_n_out, nullptr, 0, nullptr);
}

int gf_tag_alignments_low_dev(gf_ctx* ctx, const void* d_recs, size_t n, int insert_size, int sd, int clip_dist, int anchor_mapq,
                              void* d_out, size_t cap, void* d_n_out, void* d_low, size_t low_cap, void* d_n_low) {
    if (!ctx || !d_n_out || !d_n_low || (n && !d_recs) || (cap && !d_out) || (low_cap && !d_low)) return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    return launch_tag(ctx, d_recs, n, insert_size, sd, clip_dist, anchor_mapq, d_out, cap, d_n_out, d_low, low_cap, d_n_low);
}

int gf_alnrec_keys_dev(gf_ctx* ctx, const void* d_recs, size_t n, void* d_keys) {
    if (!ctx || !d_keys || (n && !d_recs)) return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    return launch_alnrec_keys(ctx, d_recs, n, d_keys);
}

int gf_tag_alignments_keys_dev(gf_ctx* ctx, const void* d_recs, const void* d_keys, size_t n, int insert_size, int sd, int clip_dist, int anchor_mapq,
                               void* d_out, size_t cap, void* d_n_out, void* d_low, size_t low_cap, void* d_n_low) {
    if (!ctx || !d_n_out || (n && (!d_recs || !d_keys)) || (cap && !d_out)) return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    return launch_tag(ctx, d_recs, n, insert_size, sd, clip_dist, anchor_mapq, d_out, cap, d_n_out, d_low, low_cap, d_n_low, d_keys);
}

int gf_tag_low_mapq_compact_dev(gf_ctx* ctx, const void* d_low, const void* d_n_low, size_t low_cap, const gf_dpos* table,
                                size_t n_rows, void* d_out, size_t cap, void* d_n_out) {
    if (!ctx || !d_n_out || !d_low || !d_n_low || (cap && !d_out) || (n_rows && !table)) return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    return launch_low_mapq(ctx, nullptr, 0, table, n_rows, d_out, cap, d_n_out, d_low, d_n_low, low_cap);
}

int gf_tag_alignments(gf_ctx* ctx, const gf_alnrec* recs, size_t n, int insert_size, int sd, int clip_dist,
                      int anchor_mapq, gf_taghit* out, size_t cap, size_t* n_out) {
    return tag_host(ctx, recs, n, out, cap, n_out, [&](void* d_in, void* d_out, void* d_n) {
        return launch_tag(ctx, d_in, n, insert_size, sd, clip_dist, anchor_mapq, d_out, cap, d_n, nullptr, 0, nullptr);
    });
}

int gf_tag_alignments_bam(gf_ctx* ctx, int insert_size, int sd, int clip_dist, int anchor_mapq, gf_taghit* out, size_t cap, size_t* n_out) {
    if (!ctx || !ctx->bam_recs.p) return ctx ? GF_E_STATE : GF_E_INVAL;
    const size_t n = ctx->bam_n_recs;
    return tag_host(ctx, nullptr, n, out, cap, n_out, [&](void* d_in, void* d_out, void* d_n) {
        return launch_tag(ctx, d_in, n, insert_size, sd, clip_dist, anchor_mapq, d_out, cap, d_n, nullptr, 0, nullptr);
    }, ctx->bam_recs.p);
}

int gf_tag_low_mapq_bam(gf_ctx* ctx, const gf_dpos* table, size_t n_rows, gf_taghit* out, size_t cap, size_t* n_out) {
    if (!ctx || !ctx->bam_recs.p) return ctx ? GF_E_STATE : GF_E_INVAL;
    if (n_rows && !table) return GF_E_INVAL;
    const size_t n = ctx->bam_n_recs;
    return tag_host(ctx, nullptr, n, out, cap, n_out, [&](void* d_in, void* d_out, void* d_n) {
        return launch_low_mapq(ctx, d_in, n, table, n_rows, d_out, cap, d_n, nullptr, nullptr, 0);
    }, ctx->bam_recs.p);
}

int gf_tag_low_mapq_dev(gf_ctx* ctx, const void* d_recs, size_t n, const gf_dpos* table, size_t n_rows, void* d_out,
                        size_t cap, void* d_n_out) {
    if (!ctx || !d_n_out || (n && !d_recs) || (cap && !d_out) || (n_rows && !table)) return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    return launch_low_mapq(ctx, d_recs, n, table, n_rows, d_out, cap, d_n_out, nullptr, nullptr, 0);
}

int gf_tag_low_mapq(gf_ctx* ctx, const gf_alnrec* recs, size_t n, const gf_dpos* table, size_t n_rows, gf_taghit* out,
                    size_t cap, size_t* n_out) {
    if (n_rows && !table) return GF_E_INVAL;
    return tag_host(ctx, recs, n, out, cap, n_out, [&](void* d_in, void* d_out, void* d_n) {
        return launch_low_mapq(ctx, d_in, n, table, n_rows, d_out, cap, d_n, nullptr, nullptr, 0);
    });
}


// ---- assembly ----------------------------------------------------------------------------------------
int gf_assemble_dev(gf_ctx* ctx, const void* d_pool, const void* d_nmask, const void* d_pool_off, size_t n_pools,
                    size_t total_reads, int read_len, int k, int kv, int min_count, int min_contig, void* d_contigs,
                    size_t contig_cap, void* d_n_contigs, void* d_seq, size_t seq_cap, void* d_seq_len, void* d_gap_error) {
    if (!ctx || !d_n_contigs || !d_seq_len || (n_pools && (!d_pool_off || !d_gap_error)) || (total_reads && !d_pool) ||
        (contig_cap && !d_contigs) || (seq_cap && !d_seq))
        return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    return launch_assemble(ctx, d_pool, d_nmask, d_pool_off, n_pools, total_reads, read_len, k, kv, min_count, min_contig,
                           d_contigs, contig_cap, d_n_contigs, d_seq, seq_cap, d_seq_len, d_gap_error, nullptr, nullptr, 0, false);
}

int gf_assemble_multi_dev(gf_ctx* ctx, const void* d_pool, const void* d_nmask, const void* d_pool_off, size_t n_pools,
                          size_t total_reads, int read_len, const int* k_list, const int* kv_list, int n_k, int min_count,
                          int min_contig, void* d_contigs, size_t contig_cap, void* d_n_contigs, void* d_seq, size_t seq_cap,
                          void* d_seq_len, void* d_gap_error) {
    if (!ctx || !d_n_contigs || !d_seq_len || (n_pools && (!d_pool_off || !d_gap_error)) || (total_reads && !d_pool) ||
        (contig_cap && !d_contigs) || (seq_cap && !d_seq) || n_k < 1 || !k_list || !kv_list)
        return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    if (n_k == 3 && !d_nmask && k_list[0] == 31 && k_list[1] == 41 && k_list[2] == 51 && kv_list[0] == 29 && kv_list[1] == 39 && kv_list[2] == 49) {
        const int rc = launch_assemble_sweep(ctx, d_pool, d_pool_off, n_pools, total_reads, read_len, min_count, min_contig, d_contigs, contig_cap,
                                             d_n_contigs, d_seq, seq_cap, d_seq_len, d_gap_error);
        if (rc != GF_E_UNSUPPORTED) return rc;
    }
    for (int i = 0; i < n_k; ++i) {   // the (k, k_velvet) loop of run_assembly (assemble_gaps.py:87-122); one contig list
        const int rc = launch_assemble(ctx, d_pool, d_nmask, d_pool_off, n_pools, total_reads, read_len, k_list[i], kv_list[i], min_count,
                                       min_contig, d_contigs, contig_cap, d_n_contigs, d_seq, seq_cap, d_seq_len, d_gap_error, nullptr,
                                       nullptr, 0, i > 0);
        if (rc) return rc;
    }
    return GF_OK;
}

int gf_assemble_last_launch(gf_ctx* ctx, int* threads_per_gap, uint32_t* to_middle, uint32_t* to_last) {
    if (!ctx || !threads_per_gap || !to_middle || !to_last) return GF_E_INVAL;
    *threads_per_gap = ctx->asm_last_threads;
    *to_middle = *to_last = 0;
    if (!ctx->counters.p || !ctx->asm_last_threads) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t c[6] = {};
    GF_HIP(ctx, hipMemcpyAsync(c, (uint32_t*)ctx->counters.p + 8, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->asm_last_split) { *to_middle = c[2]; *to_last = c[5]; }
    else *to_last = c[2];
    return GF_OK;
}

int gf_assemble(gf_ctx* ctx, const uint8_t* pool, const uint32_t* n_mask, const uint64_t* pool_off, size_t n_pools,
                int read_len, const int* k_list, const int* kv_list, int n_k, int min_count, int min_contig,
                gf_contig* contigs, size_t contig_cap, size_t* n_contigs, char* seq, size_t seq_cap, size_t* seq_len) {
    if (!ctx || !n_contigs || !seq_len || (n_pools && !pool_off) || n_k < 0 || (n_k && (!k_list || !kv_list)) ||
        (contig_cap && !contigs) || (seq_cap && !seq) || read_len <= 0)
        return GF_E_INVAL;
    *n_contigs = 0;
    *seq_len = 0;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t total = n_pools ? (size_t)pool_off[n_pools] : 0;
    if (total && !pool) return GF_E_INVAL;
    for (size_t g = 0; g < n_pools; ++g) if (pool_off[g] > pool_off[g + 1]) return GF_E_INVAL;
    const size_t rb = gf_packed_read_bytes(read_len), nmw = (size_t)(read_len + 31) / 32;
    int rc;
    // the host knows its pools: bound the per-workgroup workspace slices by the largest one
    // (... and the slices of the launch for deeper pools hold exactly the deepest pool of THIS call — neither what a device pipeline on
    //  the same context sized them for nor the option's default of 131 072 rows x 8 slices, 7-9 GB whether or not a pool needs it: ADVICE r4)
    struct MaxRows {
        gf_ctx* c; long saved, saved_big;
        MaxRows(gf_ctx* c_, long v, long big) : c(c_), saved(c_->asm_max_pool_reads), saved_big(c_->asm_big_pool_reads) {
            c->asm_max_pool_reads = v;
            c->asm_big_pool_reads = big;
        }
        ~MaxRows() { c->asm_max_pool_reads = saved; c->asm_big_pool_reads = saved_big; }
    };
    size_t max_rows = 1;
    for (size_t g = 0; g < n_pools; ++g) max_rows = std::max<size_t>(max_rows, (size_t)(pool_off[g + 1] - pool_off[g]));
    const size_t deepest = max_rows;
    if (ctx->asm_max_pool_reads > 0) max_rows = std::min<size_t>(max_rows, (size_t)ctx->asm_max_pool_reads);   // (a caller's own, lower bound stays: deeper pools take the second launch)
    MaxRows bound(ctx, (long)std::min<size_t>(max_rows, 0x3FFFFFFF), (long)std::min<size_t>(deepest, 0x1FFFFF));
    // device staging: [pool | pool_off | gap_error | counters | contigs | seq]
    const size_t b_pool = (total * rb + 63) & ~(size_t)63, b_off = ((n_pools + 1) * 8 + 63) & ~(size_t)63,
                 b_err = (n_pools * 4 + 63) & ~(size_t)63, b_ctg = (contig_cap * sizeof(gf_contig) + 63) & ~(size_t)63;
    if ((rc = ensure(ctx, ctx->stage_in, b_pool + b_off + b_err + 64 + 64))) return rc;
    if ((rc = ensure(ctx, ctx->stage_out, b_ctg + seq_cap + 64))) return rc;
    if (n_mask && (rc = ensure(ctx, ctx->stage_aux, total * nmw * 4 + 64))) return rc;
    uint8_t* d_pool = (uint8_t*)ctx->stage_in.p;
    uint8_t* d_off = d_pool + b_pool;
    uint8_t* d_err = d_off + b_off;
    uint8_t* d_cnt = d_err + b_err;  // u32 n_contigs @0, u64 seq_len @8
    uint8_t* d_ctg = (uint8_t*)ctx->stage_out.p;
    uint8_t* d_seq = d_ctg + b_ctg;
    if (total) GF_HIP(ctx, hipMemcpyAsync(d_pool, pool, total * rb, hipMemcpyHostToDevice, ctx->stream));
    if (n_pools) GF_HIP(ctx, hipMemcpyAsync(d_off, pool_off, (n_pools + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (n_mask && total) GF_HIP(ctx, hipMemcpyAsync(ctx->stage_aux.p, n_mask, total * nmw * 4, hipMemcpyHostToDevice, ctx->stream));
    size_t nc_total = 0, seq_total = 0;
    bool nospace = false;
    std::vector<gf_contig> tmp;
    std::vector<char> tseq;
    std::vector<uint32_t> errs(n_pools);
    for (int i = 0; i < n_k; ++i) {
        rc = launch_assemble(ctx, d_pool, n_mask ? ctx->stage_aux.p : nullptr, d_off, n_pools, total, read_len, k_list[i],
                             kv_list[i], min_count, min_contig, d_ctg, contig_cap, d_cnt, d_seq, seq_cap, d_cnt + 8, d_err,
                             nullptr, nullptr, 0, false);
        if (rc) return rc;
        uint32_t nc = 0;
        unsigned long long sl = 0;
        GF_HIP(ctx, hipMemcpyAsync(&nc, d_cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
        GF_HIP(ctx, hipMemcpyAsync(&sl, d_cnt + 8, 8, hipMemcpyDeviceToHost, ctx->stream));
        if (n_pools) GF_HIP(ctx, hipMemcpyAsync(errs.data(), d_err, n_pools * 4, hipMemcpyDeviceToHost, ctx->stream));
        GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t gi = 0; gi < errs.size(); ++gi)
            if (errs[gi]) {   // a pool overflowed a table or list (see the ASM_ERR_* bits in assemble.hip)
                ctx->last_error = "assembly workspace overflow in pool " + std::to_string(gi) + " (code " + std::to_string(errs[gi]) + ")";
                return GF_E_UNSUPPORTED;
            }
        if (nc > contig_cap || sl > seq_cap || nc_total + nc > contig_cap || seq_total + sl > seq_cap) {
            nospace = true;
            nc_total += nc;
            seq_total += sl;
            continue;
        }
        tmp.resize(nc);
        tseq.resize(sl);
        if (nc) GF_HIP(ctx, hipMemcpy(tmp.data(), d_ctg, nc * sizeof(gf_contig), hipMemcpyDeviceToHost));
        if (sl) GF_HIP(ctx, hipMemcpy(tseq.data(), d_seq, sl, hipMemcpyDeviceToHost));
        // deterministic order inside this (k, kv): gap, length descending, sequence ascending
        std::sort(tmp.begin(), tmp.end(), [&](const gf_contig& a, const gf_contig& b) {
            if (a.gap != b.gap) return a.gap < b.gap;
            if (a.length != b.length) return a.length > b.length;
            return memcmp(tseq.data() + a.seq_off, tseq.data() + b.seq_off, a.length) < 0;
        });
        for (const gf_contig& c : tmp) {
            gf_contig o = c;
            o.seq_off = seq_total;
            memcpy(seq + seq_total, tseq.data() + c.seq_off, c.length);
            contigs[nc_total++] = o;
            seq_total += c.length;
        }
    }
    *n_contigs = nc_total;
    *seq_len = seq_total;
    if (nospace) return GF_E_NOSPACE;
    // final order: (gap, pair index, ...) — stable sort by gap keeps the per-pair order
    std::stable_sort(contigs, contigs + nc_total, [](const gf_contig& a, const gf_contig& b) { return a.gap < b.gap; });
    return GF_OK;
}

int gf_count_kmers(gf_ctx* ctx, const uint8_t* pool, const uint32_t* n_mask, size_t n_reads, int read_len, int k,
                   int min_count, uint64_t* kmers, uint32_t* counts, size_t cap, size_t* n_out) {
    if (!ctx || !n_out || (n_reads && !pool) || (cap && (!kmers || !counts)) || read_len <= 0) return GF_E_INVAL;
    *n_out = 0;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t rb = gf_packed_read_bytes(read_len), nmw = (size_t)(read_len + 31) / 32;
    int rc;
    const size_t b_pool = (n_reads * rb + 63) & ~(size_t)63;
    if ((rc = ensure(ctx, ctx->stage_in, b_pool + 256))) return rc;
    if ((rc = ensure(ctx, ctx->stage_out, cap * 20 + 256))) return rc;
    if (n_mask && (rc = ensure(ctx, ctx->stage_aux, n_reads * nmw * 4 + 64))) return rc;
    uint8_t* d_pool = (uint8_t*)ctx->stage_in.p;
    uint8_t* d_off = d_pool + b_pool;   // 2 x u64
    uint8_t* d_err = d_off + 64;
    uint8_t* d_cnt = d_err + 64;
    uint8_t* d_keys = (uint8_t*)ctx->stage_out.p;
    uint8_t* d_counts = d_keys + ((cap * 16 + 63) & ~(size_t)63);
    const uint64_t off[2] = {0, n_reads};
    if (n_reads) GF_HIP(ctx, hipMemcpyAsync(d_pool, pool, n_reads * rb, hipMemcpyHostToDevice, ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(d_off, off, 16, hipMemcpyHostToDevice, ctx->stream));
    if (n_mask && n_reads) GF_HIP(ctx, hipMemcpyAsync(ctx->stage_aux.p, n_mask, n_reads * nmw * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_assemble(ctx, d_pool, n_mask ? ctx->stage_aux.p : nullptr, d_off, 1, n_reads, read_len, k, 0, min_count, 0,
                         nullptr, 0, d_cnt, nullptr, 0, d_cnt + 8, d_err, d_keys, d_counts, cap, false);
    if (rc) return rc;
    uint32_t n = 0, err = 0;
    GF_HIP(ctx, hipMemcpyAsync(&n, d_cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (err) return GF_E_UNSUPPORTED;
    *n_out = n;
    if (n > cap) return GF_E_NOSPACE;
    std::vector<uint64_t> hk(2 * (size_t)n);
    std::vector<uint32_t> hc(n);
    if (n) {
        GF_HIP(ctx, hipMemcpy(hk.data(), d_keys, (size_t)n * 16, hipMemcpyDeviceToHost));
        GF_HIP(ctx, hipMemcpy(hc.data(), d_counts, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return hk[2 * a] < hk[2 * b] || (hk[2 * a] == hk[2 * b] && hk[2 * a + 1] < hk[2 * b + 1]);
    });
    for (uint32_t i = 0; i < n; ++i) {
        kmers[2 * (size_t)i] = hk[2 * (size_t)order[i]];
        kmers[2 * (size_t)i + 1] = hk[2 * (size_t)order[i] + 1];
        counts[i] = hc[order[i]];
    }
    return GF_OK;
}

// ---- memory + timing helpers -------------------------------------------------------------------------
int gf_dev_alloc(gf_ctx* ctx, size_t bytes, void** d_ptr) {
    if (!ctx || !d_ptr) return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    GF_HIP(ctx, hipMalloc(d_ptr, bytes ? bytes : 1));
    return GF_OK;
}
int gf_dev_free(gf_ctx* ctx, void* d_ptr) {
    if (!ctx) return GF_E_INVAL;
    if (d_ptr) GF_HIP(ctx, hipFree(d_ptr));
    return GF_OK;
}
int gf_memcpy_h2d(gf_ctx* ctx, void* d, const void* h, size_t bytes) {
    if (!ctx) return GF_E_INVAL;
    GF_HIP(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GF_OK;
}
int gf_memcpy_d2h(gf_ctx* ctx, void* h, const void* d, size_t bytes) {
    if (!ctx) return GF_E_INVAL;
    GF_HIP(ctx, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GF_OK;
}
int gf_memset_dev(gf_ctx* ctx, void* d, int value, size_t bytes) {
    if (!ctx) return GF_E_INVAL;
    GF_HIP(ctx, hipMemsetAsync(d, value, bytes, ctx->stream));
    return GF_OK;
}

int gf_timing_enable(gf_ctx* ctx, int on) {
    if (!ctx) return GF_E_INVAL;
    ctx->timing = on != 0;
    return GF_OK;
}
int gf_timing_reset(gf_ctx* ctx) {
    if (!ctx) return GF_E_INVAL;
    (void)hipStreamSynchronize(ctx->stream);
    drain_timing(ctx);
    for (int i = 0; i < N_KERNEL_SLOTS; ++i) { ctx->t_total[i] = 0; ctx->t_count[i] = 0; }
    return GF_OK;
}
int gf_timing_read(gf_ctx* ctx, int which, double* total_ms, uint64_t* launches) {
    if (!ctx || which < 0 || which >= N_KERNEL_SLOTS) return GF_E_INVAL;
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain_timing(ctx);
    if (total_ms) *total_ms = ctx->t_total[which];
    if (launches) *launches = ctx->t_count[which];
    return GF_OK;
}

}  // extern "C"
