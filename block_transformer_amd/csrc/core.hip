// Error plumbing, version, hipGraph and hipEvent helpers of libbt_hip.so.
#include <stdarg.h>

#include <vector>

#include "bt_common.h"

namespace bt {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return BT_ERR_HIP;
}

bool g_prof_on = false;
struct ProfRec {
    int fam;
    hipEvent_t a, b;
};
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;
static size_t g_prof_used = 0;

hipEvent_t prof_event() {
    if (g_prof_used == g_prof_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        g_prof_pool.push_back(e);
    }
    return g_prof_pool[g_prof_used++];
}
void prof_push(int family, hipEvent_t a, hipEvent_t b) { g_prof_recs.push_back({family, a, b}); }

}  // namespace bt

struct bt_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};
struct bt_event {
    hipEvent_t ev;
};

extern "C" int bt_version(void) { return BT_VERSION; }
extern "C" const char* bt_last_error(void) { return bt::g_err; }

extern "C" int bt_graph_begin_capture(bt_stream stream) {
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return bt::hip_fail(e, "hipStreamBeginCapture");
    return BT_OK;
}

extern "C" int bt_graph_end_capture(bt_stream stream, bt_graph** out) {
    BT_CHECK_ARG(out, "bt_graph_end_capture: null out");
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
    if (e != hipSuccess) return bt::hip_fail(e, "hipStreamEndCapture");
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        return bt::hip_fail(e, "hipGraphInstantiate");
    }
    bt_graph* r = new bt_graph{g, x};
    *out = r;
    return BT_OK;
}

extern "C" int bt_graph_launch(bt_graph* g, bt_stream stream) {
    BT_CHECK_ARG(g, "bt_graph_launch: null graph");
    hipError_t e = hipGraphLaunch(g->exec, (hipStream_t)stream);
    if (e != hipSuccess) return bt::hip_fail(e, "hipGraphLaunch");
    return BT_OK;
}

extern "C" int bt_graph_destroy(bt_graph* g) {
    if (!g) return BT_OK;
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
    return BT_OK;
}

extern "C" int bt_event_create(bt_event** out) {
    BT_CHECK_ARG(out, "bt_event_create: null out");
    hipEvent_t ev;
    hipError_t e = hipEventCreate(&ev);
    if (e != hipSuccess) return bt::hip_fail(e, "hipEventCreate");
    *out = new bt_event{ev};
    return BT_OK;
}
extern "C" int bt_event_record(bt_event* e, bt_stream stream) {
    BT_CHECK_ARG(e, "bt_event_record: null event");
    hipError_t r = hipEventRecord(e->ev, (hipStream_t)stream);
    if (r != hipSuccess) return bt::hip_fail(r, "hipEventRecord");
    return BT_OK;
}
extern "C" int bt_event_elapsed_ms(bt_event* start, bt_event* stop, float* ms_out) {
    BT_CHECK_ARG(start && stop && ms_out, "bt_event_elapsed_ms: null argument");
    hipError_t r = hipEventSynchronize(stop->ev);
    if (r != hipSuccess) return bt::hip_fail(r, "hipEventSynchronize");
    r = hipEventElapsedTime(ms_out, start->ev, stop->ev);
    if (r != hipSuccess) return bt::hip_fail(r, "hipEventElapsedTime");
    return BT_OK;
}
extern "C" int bt_event_destroy(bt_event* e) {
    if (!e) return BT_OK;
    (void)hipEventDestroy(e->ev);
    delete e;
    return BT_OK;
}

extern "C" int bt_prof_enable(int on) {
    bt::g_prof_recs.clear();
    bt::g_prof_used = 0;
    bt::g_prof_on = on != 0;
    return BT_OK;
}

extern "C" int bt_prof_trace(double* start_ms, double* dur_ms, int32_t* family, int64_t capacity, int64_t* n_out) {
    BT_CHECK_ARG(n_out && capacity >= 0 && (capacity == 0 || (start_ms && dur_ms && family)), "bt_prof_trace: bad arguments");
    const int64_t n = (int64_t)bt::g_prof_recs.size();
    *n_out = n;
    if (capacity == 0) return BT_OK;   // size query
    BT_CHECK_ARG(capacity >= n, "bt_prof_trace: capacity below the record count (query with capacity 0 first)");
    for (int64_t i = 0; i < n; ++i) {
        const auto& r = bt::g_prof_recs[(size_t)i];
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return bt::hip_fail(e, "hipEventSynchronize");
        float t0 = 0.f, d = 0.f;
        if (i > 0) {
            e = hipEventElapsedTime(&t0, bt::g_prof_recs[0].a, r.a);
            if (e != hipSuccess) return bt::hip_fail(e, "hipEventElapsedTime");
        }
        e = hipEventElapsedTime(&d, r.a, r.b);
        if (e != hipSuccess) return bt::hip_fail(e, "hipEventElapsedTime");
        start_ms[i] = t0;
        dur_ms[i] = d;
        family[i] = r.fam;
    }
    return BT_OK;
}

extern "C" int bt_prof_collect(double* ms_out, int64_t* count_out, int n_fam) {
    BT_CHECK_ARG(ms_out && count_out && n_fam == BT_PROF_NFAM, "bt_prof_collect: bad arguments");
    for (int i = 0; i < n_fam; ++i) {
        ms_out[i] = 0.0;
        count_out[i] = 0;
    }
    for (const auto& r : bt::g_prof_recs) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return bt::hip_fail(e, "hipEventSynchronize");
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) return bt::hip_fail(e, "hipEventElapsedTime");
        ms_out[r.fam] += ms;
        count_out[r.fam] += 1;
    }
    bt::g_prof_recs.clear();
    bt::g_prof_used = 0;
    return BT_OK;
}
