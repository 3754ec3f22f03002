// Optional per-launch timing with HIP events on the stream the kernels run on (bench.py's live roofline
// measurement).  Off by default: a disabled ProfScope costs one branch.
#include <cstring>
#include <mutex>

#include "common.h"

namespace pea {
namespace {
struct Rec {
    const char *name;
    hipEvent_t beg, end;
    double units;  // algorithmic bytes (or 0) attributed to the launch
    double gathered, table;  // see ProfScope (common.h)
};
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
}  // namespace

bool prof_enabled() { return g_on; }

static hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(const char *name, hipStream_t stream, double units, double gathered, double table)
    : idx_(-1), stream_(stream) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r{name, get_event(), get_event(), units, gathered, table};
    (void)hipEventRecord(r.beg, stream);
    idx_ = (int)g_recs.size();
    g_recs.push_back(r);
}

ProfScope::~ProfScope() {
    if (idx_ < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_recs[(size_t)idx_].end, stream_);
}

}  // namespace pea

extern "C" int pea_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(pea::g_mu);
    pea::g_on = on != 0;
    return PEA_OK;
}

// Waits for every recorded launch, returns up to `max` records (name [32 bytes each], milliseconds, units)
// in launch order, and clears the log.
extern "C" int pea_profile_read_ex(int max, char *names, float *ms, double *units, double *gathered, double *table,
                                   int *count);

extern "C" int pea_profile_read(int max, char *names, float *ms, double *units, int *count) {
    return pea_profile_read_ex(max, names, ms, units, nullptr, nullptr, count);
}

extern "C" int pea_profile_read_ex(int max, char *names, float *ms, double *units, double *gathered, double *table,
                                   int *count) {
    std::lock_guard<std::mutex> lk(pea::g_mu);
    int n = 0;
    for (auto &r : pea::g_recs) {
        float t = 0.f;
        if (hipEventSynchronize(r.end) == hipSuccess) (void)hipEventElapsedTime(&t, r.beg, r.end);
        if (n < max) {
            if (names) {
                strncpy(names + (size_t)n * 32, r.name, 31);
                names[(size_t)n * 32 + 31] = 0;
            }
            if (ms) ms[n] = t;
            if (units) units[n] = r.units;
            if (gathered) gathered[n] = r.gathered;
            if (table) table[n] = r.table;
            ++n;
        }
        pea::g_pool.push_back(r.beg);
        pea::g_pool.push_back(r.end);
    }
    pea::g_recs.clear();
    if (count) *count = n;
    return PEA_OK;
}
