// Error reporting and the optional per-kernel hipEvent timing of libsplat_hip.so.
#include <stdarg.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void splat_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *splat_last_error(void) { return g_err; }
extern "C" int splat_abi_version(void) { return 21; }

#ifndef SPLAT_BUILD_ID
#define SPLAT_BUILD_ID "unstamped"
#endif
// hash of the sources this binary was built from (csrc/Makefile): measurement records carry it, bench.py refuses PMC
// constants taken from another build
extern "C" const char *splat_build_id(void) { return SPLAT_BUILD_ID; }

// Deterministic mode: every kernel this library launches for a backward is free of float atomics, so two runs
// on the same inputs give bit-identical gradients.  The frame-batch path and the per-frame operators on this package's own
// sort (pair records + quarter lists from the forward's cull words) are deterministic already; the flag closes the
// remaining doors: the block-level matrix-core kernel (its carried survivors add with atomics) gives way to the DPP pair
// kernel, and a foreign idx_sorted (atomic backward) is refused.
// (std::atomic: the flag may be flipped by one thread while another launches; set it before use for a defined kernel choice)
static std::atomic<int> g_deterministic{0};
extern "C" void splat_set_deterministic(int on) { g_deterministic.store(on != 0, std::memory_order_relaxed); }
extern "C" int splat_get_deterministic(void) { return g_deterministic.load(std::memory_order_relaxed); }
bool splat_deterministic() { return g_deterministic.load(std::memory_order_relaxed) != 0; }

// Options: kernel selection through the ABI (splat_set_option) instead of getenv() inside the library.
static const char *const g_opt_key[SPLAT_OPT_COUNT] = {"bwd_quarters", "bwd_kernel_dpp", "sets_std", "bin_slot_keys"};
static std::atomic<int> g_opt[SPLAT_OPT_COUNT] = {{1}, {0}, {1}, {0}};
int splat_option(int id) { return g_opt[id].load(std::memory_order_relaxed); }
extern "C" int splat_set_option(const char *key, int value) {
    SPLAT_CHECK_ARG(key != nullptr, "null key");
    if (strcmp(key, "deterministic") == 0) { splat_set_deterministic(value); return SPLAT_OK; }
    for (int i = 0; i < SPLAT_OPT_COUNT; ++i)
        if (strcmp(key, g_opt_key[i]) == 0) { g_opt[i].store(value, std::memory_order_relaxed); return SPLAT_OK; }
    splat_set_error("splat_set_option: unknown key '%s'", key);
    return SPLAT_E_ARG;
}
extern "C" int splat_get_option(const char *key, int *value) {
    SPLAT_CHECK_ARG(key != nullptr && value != nullptr, "null pointer");
    if (strcmp(key, "deterministic") == 0) { *value = splat_get_deterministic(); return SPLAT_OK; }
    for (int i = 0; i < SPLAT_OPT_COUNT; ++i)
        if (strcmp(key, g_opt_key[i]) == 0) { *value = splat_option(i); return SPLAT_OK; }
    splat_set_error("splat_get_option: unknown key '%s'", key);
    return SPLAT_E_ARG;
}

namespace {
struct Pending {
    std::string name;
    hipEvent_t a, b;
};
std::mutex g_mu;
bool g_on = false;
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_pool;
std::map<std::string, std::pair<double, int>> g_acc;
thread_local Pending g_cur;

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

void drain_locked() {
    for (auto &p : g_pending) {
        float ms = 0.f;
        (void)hipEventSynchronize(p.b);
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto &acc = g_acc[p.name];
            acc.first += ms;
            acc.second += 1;
        }
        g_pool.push_back(p.a);
        g_pool.push_back(p.b);
    }
    g_pending.clear();
}
}  // namespace

bool splat_profile_on() { return g_on; }

void splat_profile_begin(const char *name, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_cur.name = name;
    g_cur.a = get_event();
    g_cur.b = get_event();
    (void)hipEventRecord(g_cur.a, s);
}

void splat_profile_end(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_cur.b, s);
    g_pending.push_back(g_cur);
}

extern "C" void splat_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
}

extern "C" void splat_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
    g_acc.clear();
}

extern "C" int splat_profile_read(const char *prefix, double *total_ms, int *launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
    double t = 0.0;
    int n = 0;
    const size_t L = prefix ? strlen(prefix) : 0;
    for (auto &kv : g_acc) {
        if (L == 0 || kv.first.compare(0, L, prefix) == 0) {
            t += kv.second.first;
            n += kv.second.second;
        }
    }
    if (total_ms) *total_ms = t;
    if (launches) *launches = n;
    return SPLAT_OK;
}
