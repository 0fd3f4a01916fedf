// Error reporting + version of the C ABI (include/transfuser_hip.h).
#include "tf_common.h"
#include "../../include/transfuser_hip.h"
#include <stdarg.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include "tf_gemm_engine.h"

namespace tf {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
// TF_ABLATE=substr[,substr...]: launches of kernels whose (source) name contains one of the substrings become no-ops.  Timing diagnosis only
// ("what would the step cost if this family were free"): the results of an ablated run are garbage.
bool ablated(const char* kernel) {
    static const std::string list = [] { const char* e = getenv("TF_ABLATE"); return std::string(e ? e : ""); }();
    if (list.empty()) return false;
    size_t pos = 0;
    while (pos <= list.size()) {
        size_t end = list.find(',', pos);
        if (end == std::string::npos) end = list.size();
        if (end > pos && strstr(kernel, list.substr(pos, end - pos).c_str())) return true;
        pos = end + 1;
    }
    return false;
}
}  // namespace tf

// tf_build_id: sha256 (first 16 hex digits) of the sources this library was compiled from - transfuser_amd/csrc/*.{cpp,h} and
// include/transfuser_hip.h, in sorted order - written by transfuser_amd/build.py into the compile line of this file.  _lib.load() compares it
// with the sources next to the library and warns when a shipped .so is stale; tests/test_abi.py asserts equality after build().
#ifndef TF_BUILD_ID
#define TF_BUILD_ID "unknown"
#endif
extern "C" const char* tf_build_id(void) { return TF_BUILD_ID; }
extern "C" int tf_version(void) { return 101; }
extern "C" long tf_streamk_launches(void) { return tf::streamk_count(0); }
extern "C" const char* tf_last_error(void) { return tf::g_err; }

// ---- GEMM plan cache / autotuner switches -----------------------------------------------------------------------
namespace tf {
typedef std::tuple<std::string, int, int, int, int, int> PlanKey;
static std::map<PlanKey, GemmPlan> g_plans;
static std::mutex g_plan_mu;
static bool g_autotune = false;

bool plan_lookup(const char* what, int M, int N, int K, int batch, int acc, GemmPlan* out) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plans.find(PlanKey(what, M, N, K, batch, acc));
    if (it == g_plans.end()) return false;
    *out = it->second;
    return true;
}
void plan_store(const char* what, int M, int N, int K, int batch, int acc, const GemmPlan& p) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    g_plans[PlanKey(what, M, N, K, batch, acc)] = p;
}
bool autotune_enabled() { return g_autotune; }
long streamk_count(int add) { static long n = 0; n += add; return n; }
int device_cus() {
#ifdef TF_EMU
    return 8;                       // small on purpose: the emulated stream-K launches cut tiles on test-sized problems
#else
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus;
    }();
    return n;
#endif
}
static GemmPlan g_forced{0, 0, 0, 0, 0};
bool forced_plan(GemmPlan* out) { if (g_forced.bm == 0) return false; *out = g_forced; return true; }
}  // namespace tf

static bool plan_tile_ok(int bm, int bn, int bk, int sk, int kind) {
    if (kind != 0) { const tf::DmaKindInfo ki = tf::dma_kind_info(kind); return kind >= 1 && kind <= tf::kDmaKinds && sk >= 1 && ki.bm == bm && ki.bn == bn && ki.bk == bk; }
    return ((bm == 128 && (bn == 32 || bn == 64 || bn == 96 || bn == 128)) || (bm == 64 && (bn == 64 || bn == 128))) && (bk == 16 || bk == 32) && sk >= 1;
}
extern "C" int tf_force_dma(int kind, int splitk) {
    if (kind < 1 || kind > tf::kDmaKinds || splitk < 1) { tf::set_error("tf_force_dma: unknown LDS-DMA configuration %d", kind); return -1; }
    const tf::DmaKindInfo ki = tf::dma_kind_info(kind);
    tf::g_forced = tf::GemmPlan{ki.bm, ki.bn, ki.bk, splitk, kind};
    return 0;
}
extern "C" int tf_force_plan(int bm, int bn, int bk, int splitk) {
    const bool ok = bm == 0 || (((bm == 128 && (bn == 32 || bn == 64 || bn == 96 || bn == 128)) || (bm == 64 && (bn == 64 || bn == 128))) && (bk == 16 || bk == 32) && splitk >= 1);
    if (!ok) { tf::set_error("tf_force_plan: unsupported tiling %dx%dx%d", bm, bn, bk); return -1; }
    tf::g_forced = tf::GemmPlan{bm, bn, bk, splitk, 0};
    return 0;
}
namespace tf {
static int g_precision = 0;
int gemm_precision() { return g_precision; }
}
extern "C" int tf_set_precision(int mode) {
    if (mode < 0 || mode > 3) { tf::set_error("tf_set_precision: mode %d (0 = fp32 MFMA, 1 = bf16 MFMA with fp32 accumulate, 2 = bf16x3 split: fp32-accurate on the bf16 MFMA, 3 = fp16 MFMA with fp32 accumulate)", mode); return -1; }
    tf::g_precision = mode;
    return 0;
}
extern "C" int tf_get_precision(void) { return tf::g_precision; }
extern "C" int tf_autotune(int enable) { tf::g_autotune = enable != 0; return 0; }
extern "C" int tf_plans_count(void) { std::lock_guard<std::mutex> lk(tf::g_plan_mu); return (int)tf::g_plans.size(); }
extern "C" int tf_plans_clear(void) { std::lock_guard<std::mutex> lk(tf::g_plan_mu); tf::g_plans.clear(); return 0; }
extern "C" int tf_plans_save(const char* path) {
    std::lock_guard<std::mutex> lk(tf::g_plan_mu);
    FILE* f = fopen(path, "w");
    if (!f) { tf::set_error("tf_plans_save: cannot open %s", path); return -1; }
    fprintf(f, "# transfuser_hip GEMM plans: site;M;N;K;batch;acc;bm;bn;bk;splitk;kind   (kind 0 = register-staged kernel, >= 1 = LDS-DMA configuration; splitk >= 1000000 = deterministic two-pass split-K with splitk - 1000000 slices)\n");
    for (auto& kv : tf::g_plans)
        fprintf(f, "%s;%d;%d;%d;%d;%d;%d;%d;%d;%d;%d\n", std::get<0>(kv.first).c_str(), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first),
                std::get<4>(kv.first), std::get<5>(kv.first), kv.second.bm, kv.second.bn, kv.second.bk, kv.second.splitk, kv.second.kind);
    fclose(f);
    return 0;
}
extern "C" int tf_plans_load(const char* path) {
    FILE* f = fopen(path, "r");
    if (!f) { tf::set_error("tf_plans_load: cannot open %s", path); return -1; }
    char line[512];
    int n = 0;
    while (fgets(line, sizeof(line), f)) {
        if (line[0] == '#') continue;
        char site[256];
        int M, N, K, b, acc, bm, bn, bk, sk, kind = 0;
        char* semi = strchr(line, ';');
        if (!semi || (size_t)(semi - line) >= sizeof(site)) continue;
        memcpy(site, line, semi - line); site[semi - line] = 0;
        if (sscanf(semi + 1, "%d;%d;%d;%d;%d;%d;%d;%d;%d;%d", &M, &N, &K, &b, &acc, &bm, &bn, &bk, &sk, &kind) < 9) continue;   // 10th field optional (r01 files)
        if (!plan_tile_ok(bm, bn, bk, sk, kind)) continue;
        tf::plan_store(site, M, N, K, b, acc, tf::GemmPlan{bm, bn, bk, sk, kind});
        ++n;
    }
    fclose(f);
    return n;
}
