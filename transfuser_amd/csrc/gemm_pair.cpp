// tf_gemm_pair_begin / tf_gemm_pair_end: two independent tf_gemm_f32 calls in ONE grid (tf_gemm_engine.h, "pair launch").
// The reference's autograd issues the weight gradient and the input gradient of a layer as two cuBLAS / cuDNN launches on one stream
// (torch.nn.Conv2d / Linear backward behind transfuser.py:380,442,545-549); here both read dy, write disjoint outputs and share a launch.
#include "tf_gemm_engine.h"
#include "../../include/transfuser_hip.h"

namespace tf {
namespace {
struct Held { PlainOp la, lb; GemmEpi ep; int M, N, K; bool a_kc, b_kc; GemmPlan p; const char* what; };
thread_local bool g_active = false;
thread_local int g_n = 0;
thread_local Held g_held[2];
thread_local long g_pairs = 0, g_singles = 0;

PairSide side_of(const Held& h) {
    const CfgGeom cg = cfg_geom(h.ep, h.M, h.N, h.K, h.p.splitk, 64, 64, h.p.bk, 2);
    PairSide s;
    s.la = h.la; s.lb = h.lb; s.ep = cg.epg; s.M = h.M; s.N = h.N; s.K = h.K;
    s.tiles_m = cg.tiles_m; s.tiles_n = cg.tiles_n; s.kchunk = cg.kchunk; s.gx = cg.tiles_m * cg.tiles_n; s.gy = cg.nsplit;
    return s;
}

// a held call is always a 64 x 64 register-staged plan (tf_gemm_engine.h:launch_gemm): the same launch launch_plan would have issued
template <bool A_KC, bool B_KC>
void launch_single(const Held& h, void* stream) {
    if (h.p.bk == 32) launch_cfg<64, 64, 2, 32, PlainOp, A_KC, PlainOp, B_KC>(h.la, h.lb, h.ep, h.M, h.N, h.K, 1, h.p.splitk, stream);
    else launch_cfg<64, 64, 2, 16, PlainOp, A_KC, PlainOp, B_KC>(h.la, h.lb, h.ep, h.M, h.N, h.K, 1, h.p.splitk, stream);
}

void launch_alone(const Held& h, void* stream) {
    if (h.a_kc) { if (h.b_kc) launch_single<true, true>(h, stream); else launch_single<true, false>(h, stream); }
    else { if (h.b_kc) launch_single<false, true>(h, stream); else launch_single<false, false>(h, stream); }
    ++g_singles;
}

// first = the weight gradient (operands [k][m] / [k][n]: both "IC"), second = the input gradient (A [m][k], B [k][n]); the longer k-chains go first
template <int BK1, int BK2>
void launch_wgrad_dgrad(const Held& w, const Held& d, void* stream) {
    const PairSide s1 = side_of(w), s2 = side_of(d);
    const int n1 = s1.gx * s1.gy, n2 = s2.gx * s2.gy, n1pad = (n1 + 7) & ~7;
    TF_LAUNCH((gemm_pair_kernel<BK1, false, false, BK2, true, false>), dim3(n1pad + n2), dim3(256), stream, s1, s2, n1, n1pad);
    ++g_pairs;
}
}  // namespace

bool pair_capturing() { return g_active && g_n < 2; }

bool pair_hold(const PlainOp& la, const PlainOp& lb, const GemmEpi& ep, int M, int N, int K, bool a_kc, bool b_kc, const GemmPlan& p, const char* what) {
    if (!pair_capturing()) return false;
    Held& h = g_held[g_n++];
    h.la = la; h.lb = lb; h.ep = ep; h.M = M; h.N = N; h.K = K; h.a_kc = a_kc; h.b_kc = b_kc; h.p = p; h.what = what;
    return true;
}
}  // namespace tf

using namespace tf;

extern "C" int tf_gemm_pair_begin(void) {
    TF_REQUIRE(!g_active, "tf_gemm_pair_begin: a pair is already open on this thread");
    g_active = true; g_n = 0;
    return 0;
}

extern "C" int tf_gemm_pair_end(void* stream) {
    TF_REQUIRE(g_active, "tf_gemm_pair_end without tf_gemm_pair_begin");
    g_active = false;
    const int n = g_n;
    g_n = 0;
    if (n == 2) {
        // which of the two is the weight gradient ([tn]: neither operand k-contiguous) and which the input gradient ([nn])?
        const Held* w = nullptr; const Held* d = nullptr;
        for (int i = 0; i < 2; ++i) {
            if (!g_held[i].a_kc && !g_held[i].b_kc && !w) w = &g_held[i];
            else if (g_held[i].a_kc && !g_held[i].b_kc && !d) d = &g_held[i];
        }
        if (w && d) {
            if (w->p.bk == 32) { if (d->p.bk == 32) launch_wgrad_dgrad<32, 32>(*w, *d, stream); else launch_wgrad_dgrad<32, 16>(*w, *d, stream); }
            else { if (d->p.bk == 32) launch_wgrad_dgrad<16, 32>(*w, *d, stream); else launch_wgrad_dgrad<16, 16>(*w, *d, stream); }
            return launch_status("tf_gemm_pair_end");
        }
    }
    for (int i = 0; i < n; ++i) launch_alone(g_held[i], stream);      // one held call, or a layout pair without a joint kernel: in call order
    return launch_status("tf_gemm_pair_end");
}

// pairs / single launches issued through tf_gemm_pair_end so far on this thread (tests assert that the joint kernel really ran)
extern "C" long tf_gemm_pair_count(int singles) { return singles ? g_singles : g_pairs; }
