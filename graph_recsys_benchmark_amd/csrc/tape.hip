// Launch tape: record the kernel launches of a call sequence once, replay them with no host logic in between.
// See common.h (PEA_LAUNCH).  Recording is per thread; replay may come from any thread.
#include "common.h"

namespace pea {
namespace {
thread_local Tape *g_tape = nullptr;
}
Tape *tape_active() { return g_tape; }
}  // namespace pea

struct pea_tape {
    pea::Tape tape;
};

extern "C" int pea_tape_create(pea_tape **out) {
    PEA_REQUIRE(out != nullptr, PEA_ERR_ARG, "tape: out is null");
    *out = new pea_tape();
    return PEA_OK;
}

extern "C" int pea_tape_destroy(pea_tape *t) {
    if (t && pea::g_tape == &t->tape) pea::g_tape = nullptr;
    delete t;
    return PEA_OK;
}

// Starts (re)recording: the tape is cleared, and every launch the library issues from this thread until pea_tape_end is
// executed AND recorded.  Profiling (pea_profile_enable) must be off: its events are not part of a tape.
extern "C" int pea_tape_begin(pea_tape *t) {
    PEA_REQUIRE(t != nullptr, PEA_ERR_ARG, "tape: null tape");
    PEA_REQUIRE(pea::g_tape == nullptr, PEA_ERR_ARG, "tape: another tape is recording on this thread");
    PEA_REQUIRE(!pea::prof_enabled(), PEA_ERR_ARG, "tape: per-launch profiling is on (events are not recorded)");
    t->tape.ops.clear();
    pea::g_tape = &t->tape;
    return PEA_OK;
}

extern "C" int pea_tape_end(pea_tape *t) {
    PEA_REQUIRE(t != nullptr && pea::g_tape == &t->tape, PEA_ERR_ARG, "tape: this tape is not recording");
    pea::g_tape = nullptr;
    return PEA_OK;
}

extern "C" int pea_tape_length(const pea_tape *t) { return t ? (int)t->tape.ops.size() : 0; }

extern "C" int pea_tape_replay(const pea_tape *t, void *stream) {
    PEA_REQUIRE(t != nullptr, PEA_ERR_ARG, "tape: null tape");
    PEA_REQUIRE(pea::g_tape == nullptr, PEA_ERR_ARG, "tape: replay while a tape is recording on this thread");
    for (const auto &op : t->tape.ops) op((hipStream_t)stream);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}
