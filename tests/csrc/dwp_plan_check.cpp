// Host-side check of the work plan of the weight-gradient GEMM (mirror_nerf_amd/csrc/mnrf_dwp.h), compiled with g++ by
// tests/test_dwp_plan_cpu.py: the same inline functions the kernel and its launcher use.
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "mnrf_dwp.h"

using namespace mnrf;

static int check(const DwpPlan& p) {
    const int NV = DWP_JOBS * p.n_eval;
    std::set<int> slots;
    long long P = 0;
    for (int j = 0; j < DWP_JOBS; ++j) {
        const int w = dwp_weight(j);
        for (int e = 0; e < p.n_eval; ++e) {
            const int v = j * p.n_eval + e, n = dwp_stages(p, j, e);
            std::vector<int> owner(n, -1);
            int g_lo = -1, g_hi = -1;
            for (int g = 0; g < p.G; ++g) {
                long long c0, c1;
                dwp_interval(p, g, c0, c1);
                int s_lo, s_hi;
                dwp_segment(c0, c1, P, w, n, s_lo, s_hi);
                if (s_lo < 0 || s_hi > n || s_hi < s_lo) { std::printf("bad segment g=%d v=%d [%d,%d) n=%d\n", g, v, s_lo, s_hi, n); return 1; }
                if (s_hi == s_lo) continue;
                if (g_lo >= 0 && g != g_hi + 1) { std::printf("owners of v=%d not contiguous: %d after %d\n", v, g, g_hi); return 1; }
                if (g_lo < 0) g_lo = g;
                g_hi = g;
                for (int s = s_lo; s < s_hi; ++s) {
                    if (owner[s] != -1) { std::printf("stage %d of v=%d owned twice\n", s, v); return 1; }
                    // ownership rule: the stage STARTS inside the owner's interval
                    const long long x = P + (long long)s * w;
                    if (x < c0 || x >= c1) { std::printf("stage %d of v=%d outside its owner's interval\n", s, v); return 1; }
                    owner[s] = g;
                }
                const int slot = g + v;
                if (slot >= p.G + NV) { std::printf("slot %d out of range\n", slot); return 1; }
                if (!slots.insert(slot).second) { std::printf("slot %d used twice\n", slot); return 1; }
            }
            for (int s = 0; s < n; ++s)
                if (owner[s] < 0) { std::printf("stage %d of v=%d has no owner (G=%d T=%lld)\n", s, v, p.G, p.T); return 1; }
            // the closed form the launcher and the device-side plan (dwp_plan_kernel) use for the finish kernel's owner ranges
            if (n > 0 && (dwp_owner(p, P) != g_lo || dwp_owner(p, P + (long long)(n - 1) * w) != g_hi)) {
                std::printf("dwp_owner of v=%d: [%d, %d] but the search says [%d, %d]\n", v, dwp_owner(p, P),
                            dwp_owner(p, P + (long long)(n - 1) * w), g_lo, g_hi);
                return 1;
            }
            P += (long long)n * w;
        }
    }
    if (P != p.T) { std::printf("cost line %lld != T %lld\n", P, p.T); return 1; }
    return 0;
}

int main() {
    // job table sanity: sections inside the plane layouts, weights as documented
    int wsum = 0;
    for (int j = 0; j < DWP_JOBS; ++j) {
        const DwpJob jb = dwp_job(j);
        if (jb.ya < 0 || jb.ya + jb.na > PLY_FB || jb.xa < 0 || jb.xa + jb.nx > PLX_FB || jb.na > 16 || jb.nx > 16) { std::printf("job %d out of the layout\n", j); return 1; }
        wsum += dwp_weight(j);
    }
    if (wsum != 812 + DWP_JOBS * DWP_STAGE_KIB) { std::printf("weights sum %d\n", wsum); return 1; }
    for (int j = 0; j < DWP_JOBS; ++j) {      // second-order planes: their own (shorter) layouts
        if (!dwp_has(1, j)) continue;
        const DwpJob jb = dwp_job_of(1, j);
        if (jb.bias || jb.ya + jb.na > PL2Y_FB || jb.xa + jb.nx > PL2X_FB) { std::printf("second-order job %d out of the layout\n", j); return 1; }
    }
    std::srand(7);
    int cases = 0;
    for (int it = 0; it < 1000; ++it) {
        DwpPlan p;
        p.n_eval = 1 + std::rand() % DWP_MAX_EVAL;
        for (int e = 0; e < DWP_MAX_EVAL; ++e) { p.n_sb[e] = 0; p.kind[e] = 0; }
        for (int e = 0; e < p.n_eval; ++e) {
            p.kind[e] = (std::rand() % 3) == 0;      // a third of the evaluations are second-order planes (trunk + sigma jobs only)
            const int kind = std::rand() % 4;
            p.n_sb[e] = kind == 0 ? 0 : (kind == 1 ? 4 * (1 + std::rand() % 3) : (kind == 2 ? 4 * (std::rand() % 600) : 4 * (std::rand() % 60000)));
        }
        p.T = dwp_total(p);
        if (p.T == 0) continue;
        const int cus[4] = {256, 304, 1, 64};
        p.G = dwp_pick_G(p.T, cus[it % 4]);
        if (check(p)) { std::printf("FAILED plan: n_eval=%d G=%d T=%lld\n", p.n_eval, p.G, p.T); return 1; }
        ++cases;
    }
    // the training bench's shape: 1024 + 257 rays x 64 (coarse) samples
    DwpPlan p;
    p.n_eval = 2;
    for (int e = 0; e < DWP_MAX_EVAL; ++e) { p.n_sb[e] = 0; p.kind[e] = 0; }
    p.n_sb[0] = 2048; p.n_sb[1] = 516;
    p.T = dwp_total(p);
    p.G = dwp_pick_G(p.T, 256);
    if (check(p)) return 1;
    std::printf("ok %d plans; bench plan: G=%d T=%lld (%.0f KiB per workgroup)\n", cases, p.G, p.T, (double)p.T / p.G);
    return 0;
}
