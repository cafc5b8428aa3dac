// Native executor of libfrido_hip: runs / captures "programs" (arrays of tagged op descriptors) on a
// HIP stream, owns the hipGraph objects of the captured per-step bodies, HIP-event timing on the
// launch stream, and error reporting.  One program = one U-Net forward + sampler update (or the
// VQGAN decode); the Python host builds it once per (model, batch, stage) and replays it.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <mutex>
#include <vector>

#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

void frido_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int frido_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        frido_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return FRIDO_EHIP;
    }
    return FRIDO_OK;
}

namespace {

int run_one(const FridoOp& op, frido_stream_t s) {
    switch (op.kind) {
        case FRIDO_OP_GEMM: return frido_gemm(&op.u.gemm, s);
        case FRIDO_OP_GN_STATS: return frido_gn_stats(&op.u.gn_stats, s);
        case FRIDO_OP_GN_APPLY: return frido_gn_apply(&op.u.gn_apply, s);
        case FRIDO_OP_LAYERNORM: return frido_layernorm(&op.u.layernorm, s);
        case FRIDO_OP_SOFTMAX: return frido_softmax(&op.u.softmax, s);
        case FRIDO_OP_GEGLU: return frido_geglu(&op.u.geglu, s);
        case FRIDO_OP_PACK: return frido_pack(&op.u.pack, s);
        case FRIDO_OP_RELAYOUT: return frido_relayout(&op.u.relayout, s);
        case FRIDO_OP_VQ: return frido_vq(&op.u.vq, s);
        case FRIDO_OP_SAMPLER_STEP: return frido_sampler_step(&op.u.sampler_step, s);
        case FRIDO_OP_HANDOFF: return frido_handoff(&op.u.handoff, s);
        case FRIDO_OP_RANDN: return frido_randn(&op.u.randn, s);
        case FRIDO_OP_STEP_ADD: return frido_step_add(&op.u.step_add, s);
        case FRIDO_OP_FILL: return frido_fill(&op.u.fill, s);
        case FRIDO_OP_TIME_EMB: return frido_time_emb(&op.u.time_emb, s);
        case FRIDO_OP_CONVT: return frido_convt(&op.u.convt, s);
        case FRIDO_OP_PLACE: return frido_place(&op.u.place, s);
        case FRIDO_OP_EMBED: return frido_embed(&op.u.embed, s);
        case FRIDO_OP_TO_U8: return frido_to_u8(&op.u.to_u8, s);
        case FRIDO_OP_ATTN_SMALL: return frido_attn_small(&op.u.attn_small, s);
        case FRIDO_OP_GN_FUSED: return frido_gn_fused(&op.u.gn_apply, s);
        case FRIDO_OP_COPY: return frido_copy(&op.u.copy, s);
        case FRIDO_OP_ATTN_FLASH: return frido_attn_flash(&op.u.attn_small, s);
        case FRIDO_OP_L2NORM: return frido_l2norm(&op.u.l2norm, s);
        default:
            frido_set_error("frido_run: unknown op kind %d", op.kind);
            return FRIDO_EINVAL;
    }
}

struct Graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

// side stream + a small ring of ordering events per caller stream (created on first use, never destroyed: a handful per process)
struct Side {
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    int next = 0;
};
std::mutex g_side_mu;
std::vector<std::pair<hipStream_t, Side*>> g_sides;

Side* side_of(hipStream_t main) {
    std::lock_guard<std::mutex> lk(g_side_mu);
    for (auto& p : g_sides)
        if (p.first == main) return p.second;
    Side* sd = new Side();
    if (hipStreamCreateWithFlags(&sd->stream, hipStreamNonBlocking) != hipSuccess) {
        delete sd;
        return nullptr;
    }
    for (auto& e : sd->ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    g_sides.emplace_back(main, sd);
    return sd;
}

int run_sync(const FridoSync& d, hipStream_t main, bool serial) {
    if (serial || d.from == d.to) return FRIDO_OK;       // everything on one stream: already ordered
    Side* sd = side_of(main);
    if (!sd) {
        frido_set_error("frido_run: cannot create the side stream");
        return FRIDO_EHIP;
    }
    hipStream_t from = d.from ? sd->stream : main, to = d.to ? sd->stream : main;
    hipEvent_t ev = sd->ev[sd->next];
    sd->next = (sd->next + 1) & 7;
    if (hipEventRecord(ev, from) != hipSuccess || hipStreamWaitEvent(to, ev, 0) != hipSuccess) {
        frido_set_error("frido_run: cross-stream ordering failed: %s", hipGetErrorString(hipGetLastError()));
        return FRIDO_EHIP;
    }
    return FRIDO_OK;
}

int run_prog(const FridoOp* ops, int32_t n, frido_stream_t s, bool serial) {
    for (int32_t i = 0; i < n; ++i) {
        int rc;
        if (ops[i].kind == FRIDO_OP_SYNC) {
            rc = run_sync(ops[i].u.sync, (hipStream_t)s, serial);
        } else {
            frido_stream_t st = s;
            if (ops[i].stream == 1 && !serial) {
                Side* sd = side_of((hipStream_t)s);
                if (!sd) {
                    frido_set_error("frido_run: cannot create the side stream");
                    return FRIDO_EHIP;
                }
                st = (frido_stream_t)sd->stream;
            }
            rc = run_one(ops[i], st);
        }
        if (rc != FRIDO_OK) {
            char tmp[400];
            snprintf(tmp, sizeof(tmp), "%s", g_err);
            frido_set_error("op %d (kind %d): %s", i, ops[i].kind, tmp);
            return rc;
        }
    }
    return FRIDO_OK;
}

}  // namespace

extern "C" int frido_run(const FridoOp* ops, int32_t n, frido_stream_t s) {
    if (!ops || n < 0) {
        frido_set_error("frido_run: bad arguments");
        return FRIDO_EINVAL;
    }
    static const bool serial = getenv("FRIDO_SERIAL") && atoi(getenv("FRIDO_SERIAL")) != 0;     // A/B: ignore the side stream
    return run_prog(ops, n, s, serial);
}

extern "C" int frido_run_timed(const FridoOp* ops, int32_t n, frido_stream_t s, float* ms) {
    if (!ops || n <= 0 || !ms) {
        frido_set_error("frido_run_timed: bad arguments");
        return FRIDO_EINVAL;
    }
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) return FRIDO_EHIP;
    int rc = FRIDO_OK;
    (void)hipEventRecord(ev[0], (hipStream_t)s);
    for (int32_t i = 0; i < n && rc == FRIDO_OK; ++i) {
        if (ops[i].kind != FRIDO_OP_SYNC) rc = run_one(ops[i], s);          // per-op timing: everything on the caller's stream
        (void)hipEventRecord(ev[i + 1], (hipStream_t)s);
    }
    if (rc == FRIDO_OK) {
        (void)hipEventSynchronize(ev[n]);
        for (int32_t i = 0; i < n; ++i)
            if (hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]) != hipSuccess) rc = FRIDO_EHIP;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

extern "C" int frido_graph_capture(const FridoOp* ops, int32_t n, frido_stream_t s, void** out) {
    if (!ops || n <= 0 || !out) {
        frido_set_error("frido_graph_capture: bad arguments");
        return FRIDO_EINVAL;
    }
    hipStream_t st = (hipStream_t)s;
    hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        frido_set_error("hipStreamBeginCapture: %s", hipGetErrorString(e));
        return FRIDO_EHIP;
    }
    int rc = frido_run(ops, n, s);
    Graph* g = new Graph();
    e = hipStreamEndCapture(st, &g->graph);
    if (rc != FRIDO_OK || e != hipSuccess) {
        if (e != hipSuccess) frido_set_error("hipStreamEndCapture: %s", hipGetErrorString(e));
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
        return rc != FRIDO_OK ? rc : FRIDO_EHIP;
    }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        frido_set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
        (void)hipGraphDestroy(g->graph);
        delete g;
        return FRIDO_EHIP;
    }
    *out = g;
    return FRIDO_OK;
}

extern "C" int frido_graph_launch(void* graph, frido_stream_t s) {
    Graph* g = (Graph*)graph;
    if (!g || !g->exec) {
        frido_set_error("frido_graph_launch: null graph");
        return FRIDO_EINVAL;
    }
    hipError_t e = hipGraphLaunch(g->exec, (hipStream_t)s);
    if (e != hipSuccess) {
        frido_set_error("hipGraphLaunch: %s", hipGetErrorString(e));
        return FRIDO_EHIP;
    }
    return FRIDO_OK;
}

extern "C" int frido_graph_destroy(void* graph) {
    Graph* g = (Graph*)graph;
    if (!g) return FRIDO_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return FRIDO_OK;
}

extern "C" int frido_event_create(void** ev) {
    hipEvent_t e;
    if (!ev || hipEventCreate(&e) != hipSuccess) {
        frido_set_error("hipEventCreate failed");
        return FRIDO_EHIP;
    }
    *ev = (void*)e;
    return FRIDO_OK;
}
extern "C" int frido_event_record(void* ev, frido_stream_t s) {
    if (hipEventRecord((hipEvent_t)ev, (hipStream_t)s) != hipSuccess) {
        frido_set_error("hipEventRecord failed");
        return FRIDO_EHIP;
    }
    return FRIDO_OK;
}
extern "C" int frido_event_elapsed_ms(void* start, void* stop, float* ms) {
    if (!ms) return FRIDO_EINVAL;
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess || hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) {
        frido_set_error("hipEventElapsedTime failed");
        return FRIDO_EHIP;
    }
    return FRIDO_OK;
}
extern "C" int frido_event_destroy(void* ev) {
    if (ev) (void)hipEventDestroy((hipEvent_t)ev);
    return FRIDO_OK;
}

int frido_igemm_init();
extern "C" int frido_init(void) {
    static std::once_flag once;
    static int rc = FRIDO_OK;
    std::call_once(once, [] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
            frido_set_error("frido_init: no HIP device");
            rc = FRIDO_EHIP;
            return;
        }
        rc = frido_igemm_init();
    });
    return rc;
}

// 2 (r03/r04): FridoGemm gained sk_mode / gn_part (mid-struct), FridoAttnSmall / FridoSoftmax / FridoGnStats grew, the split-K
// workspace starts with a 64-KiB ticket header the caller zeroes, two-plane operands are fp16 pairs (frido_x3_plane_format)
extern "C" int frido_abi_version(void) { return FRIDO_ABI_VERSION; }
extern "C" int frido_x3_plane_format(void) { return FRIDO_X3_F16 ? 1 : 0; }
extern "C" int frido_sizeof_op(void) { return (int)sizeof(FridoOp); }
extern "C" int frido_sizeof_desc(int32_t kind) {
    switch (kind) {
        case FRIDO_OP_GEMM: return sizeof(FridoGemm);
        case FRIDO_OP_GN_STATS: return sizeof(FridoGnStats);
        case FRIDO_OP_GN_APPLY: return sizeof(FridoGnApply);
        case FRIDO_OP_LAYERNORM: return sizeof(FridoLayerNorm);
        case FRIDO_OP_SOFTMAX: return sizeof(FridoSoftmax);
        case FRIDO_OP_GEGLU: return sizeof(FridoGeglu);
        case FRIDO_OP_PACK: return sizeof(FridoPack);
        case FRIDO_OP_RELAYOUT: return sizeof(FridoRelayout);
        case FRIDO_OP_VQ: return sizeof(FridoVq);
        case FRIDO_OP_SAMPLER_STEP: return sizeof(FridoSamplerStep);
        case FRIDO_OP_HANDOFF: return sizeof(FridoHandoff);
        case FRIDO_OP_RANDN: return sizeof(FridoRandn);
        case FRIDO_OP_STEP_ADD: return sizeof(FridoStepAdd);
        case FRIDO_OP_FILL: return sizeof(FridoFill);
        case FRIDO_OP_TIME_EMB: return sizeof(FridoTimeEmb);
        case FRIDO_OP_CONVT: return sizeof(FridoConvT);
        case FRIDO_OP_PLACE: return sizeof(FridoPlace);
        case FRIDO_OP_EMBED: return sizeof(FridoEmbed);
        case FRIDO_OP_TO_U8: return sizeof(FridoToU8);
        case FRIDO_OP_ATTN_SMALL: return sizeof(FridoAttnSmall);
        case FRIDO_OP_GN_FUSED: return sizeof(FridoGnApply);
        case FRIDO_OP_COPY: return sizeof(FridoCopy);
        case FRIDO_OP_ATTN_FLASH: return sizeof(FridoAttnSmall);
        case FRIDO_OP_SYNC: return sizeof(FridoSync);
        case FRIDO_OP_L2NORM: return sizeof(FridoL2Norm);
        default: return -1;
    }
}
extern "C" const char* frido_last_error(void) { return g_err; }

// ---- sticky numerics status (common.h): one device word per translation unit, registered at load time ----
namespace {
std::vector<frido_status_accessor>& status_words() {
    static std::vector<frido_status_accessor> v;
    return v;
}
std::vector<frido_status_address>& status_addrs() {
    static std::vector<frido_status_address> v;
    return v;
}
// frido_status_poll: ONE small kernel ORs (and optionally clears) the per-file words in stream order and leaves the result where a
// pinned host word picks it up -- one stream synchronisation instead of a device drain plus two blocking symbol copies per file
__global__ void status_gather_kernel(unsigned* const* words, int n, unsigned* out, int clear) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned all = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned w = clear ? atomicExch(words[i], 0u) : atomicOr(words[i], 0u);
        all |= w;
    }
    *out = all;
}
struct StatusPoll {
    std::mutex mu;
    unsigned** d_words = nullptr;      // device array of the per-file word addresses
    unsigned* d_out = nullptr;
    unsigned* h_out = nullptr;         // pinned
    int n = 0;
    int dev = -1;
};
StatusPoll g_poll;
}  // namespace
void frido_register_status_word(frido_status_accessor fn, frido_status_address addr) {
    status_words().push_back(fn);
    status_addrs().push_back(addr);
}

extern "C" int frido_status_poll(frido_stream_t stream, uint32_t* flags, int32_t clear) {
    if (!flags) {
        frido_set_error("frido_status_poll: null pointer");
        return FRIDO_EINVAL;
    }
    std::lock_guard<std::mutex> lk(g_poll.mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        frido_set_error("frido_status_poll: no HIP device");
        return FRIDO_EHIP;
    }
    if (g_poll.dev != dev) {           // first call (or another device): gather the word addresses of this device's copy of every file's global
        std::vector<unsigned*> host;
        for (auto fn : status_addrs()) {
            unsigned* p = fn();
            if (!p) {
                frido_set_error("frido_status_poll: hipGetSymbolAddress failed: %s", hipGetErrorString(hipGetLastError()));
                return FRIDO_EHIP;
            }
            host.push_back(p);
        }
        if (g_poll.d_words) { (void)hipFree(g_poll.d_words); (void)hipFree(g_poll.d_out); (void)hipHostFree(g_poll.h_out); g_poll.d_words = nullptr; }
        if (hipMalloc((void**)&g_poll.d_words, host.size() * sizeof(unsigned*)) != hipSuccess || hipMalloc((void**)&g_poll.d_out, sizeof(unsigned)) != hipSuccess ||
            hipHostMalloc((void**)&g_poll.h_out, sizeof(unsigned), hipHostMallocDefault) != hipSuccess ||
            hipMemcpy(g_poll.d_words, host.data(), host.size() * sizeof(unsigned*), hipMemcpyHostToDevice) != hipSuccess) {
            frido_set_error("frido_status_poll: allocation failed: %s", hipGetErrorString(hipGetLastError()));
            return FRIDO_EHIP;
        }
        g_poll.n = (int)host.size();
        g_poll.dev = dev;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(status_gather_kernel, dim3(1), dim3(64), 0, s, g_poll.d_words, g_poll.n, g_poll.d_out, (int)clear);
    if (hipMemcpyAsync(g_poll.h_out, g_poll.d_out, sizeof(unsigned), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        frido_set_error("frido_status_poll: %s", hipGetErrorString(hipGetLastError()));
        return FRIDO_EHIP;
    }
    *flags = *g_poll.h_out;
    return FRIDO_OK;
}

// diagnostic: the word of ONE translation unit (registration = link order: igemm, convgn, norm, misc, attn, flash, runtime); -1 past the end
extern "C" int frido_status_word_of(int32_t idx, uint32_t* word) {
    if (idx < 0 || idx >= (int)status_words().size() || !word) return -1;
    unsigned w = 0;
    if (status_words()[idx](&w, 0) != FRIDO_OK) return FRIDO_EHIP;
    *word = w;
    return FRIDO_OK;
}

extern "C" int frido_status_flags(uint32_t* flags, int32_t clear) {
    if (!flags) {
        frido_set_error("frido_status_flags: null pointer");
        return FRIDO_EINVAL;
    }
    // (r06, advisor) the per-file words are read with null-stream symbol copies, which do NOT wait for kernels on non-blocking streams
    // (how torch creates its side streams): drain the device first, so that the word read -- and cleared -- is the word of everything
    // launched before this call
    if (hipDeviceSynchronize() != hipSuccess) {
        frido_set_error("frido_status_flags: hipDeviceSynchronize failed: %s", hipGetErrorString(hipGetLastError()));
        return FRIDO_EHIP;
    }
    unsigned all = 0;
    for (auto fn : status_words()) {
        unsigned w = 0;
        if (fn(&w, clear) != FRIDO_OK) {
            frido_set_error("frido_status_flags: cannot read the device status word: %s", hipGetErrorString(hipGetLastError()));
            return FRIDO_EHIP;
        }
        all |= w;
    }
    *flags = all;
    return FRIDO_OK;
}

extern "C" int frido_device_info(int32_t* cu_count, int32_t* is_gfx950, int64_t* hbm_bytes) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        frido_set_error("no HIP device");
        return FRIDO_EHIP;
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (is_gfx950) *is_gfx950 = (strncmp(p.gcnArchName, "gfx950", 6) == 0) ? 1 : 0;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return FRIDO_OK;
}
